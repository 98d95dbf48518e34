# usage (GPU box): bash scripts/gpu_r3p.sh TAG -- HBM traffic of K1 / K2 on this tree's kernel sources (profiles/pmc_traffic.json), all GPU tests, the aligned-read campaign
TAG=${1:-r3p}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT/traffic/pairs_nib $OUT/traffic/pairs $OUT/traffic/reads
cd /tmp && export TMPDIR=/tmp
for LAY in pairs_nib pairs reads; do
  case $LAY in pairs_nib) ARG="--packed";; pairs) ARG="--pairs";; reads) ARG="";; esac
  i=0
  for SET in "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --pmc $SET -d $OUT/traffic/$LAY/p$i -o p$i --output-format csv -- python $R/scripts/prof_kernels.py --reads 20000000 --reps 2 $ARG > $OUT/traffic/$LAY/p$i.log 2>&1
    rc=$?; echo "traffic $LAY pass $i ($SET) rc=$rc"
    [ $rc -eq 0 ] || exit $rc
  done
done
python $R/scripts/pmc_traffic_json.py $OUT/traffic 20000000 > $OUT/pmc_traffic.json && cp $OUT/pmc_traffic.json $R/profiles/pmc_traffic.json
grep -o '"hbm_bytes_per_base": [0-9.]*' $OUT/pmc_traffic.json | tr '\n' ' '; echo
cd $R
timeout -k 10 900 python -m pytest tests -q -m gpu -x > gpurun_out/gputests_$TAG.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -6 gpurun_out/gputests_$TAG.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tests/tools/fuzz_gpu_aligned.py --seconds 150 --seed 77 > gpurun_out/fuzz_gpu_aligned_$TAG.log 2>&1
rc=$?; echo "fuzz aligned rc=$rc"; tail -3 gpurun_out/fuzz_gpu_aligned_$TAG.log
exit $rc
