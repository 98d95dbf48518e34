# usage (GPU box): bash scripts/gpu_tilekb.sh -- the largest pitch-narrowed LUT the short-lived K2 still takes on one-read-per-row planes
R=$GRAFT_REPO_ROOT
cd $R
for ROUND in 1 2; do
for KB in 32 40 46 52; do
echo "--- LUT limit $KB KB"
KBBQ_K2_TILE_LUT_KB=$KB timeout -k 10 300 python scripts/time_mixed.py 2>&1 | tail -1 | cut -c230-330,520-640
done
done
