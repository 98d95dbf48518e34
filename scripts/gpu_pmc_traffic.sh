# usage (GPU box): bash scripts/gpu_pmc_traffic.sh TAG [--pairs] -- HBM traffic of K1/K2: FETCH_SIZE and WRITE_SIZE, each in its own pass
TAG=${1:-traffic}; shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $SET -d $OUT/p$i -o p$i --output-format csv -- python $R/scripts/prof_kernels.py --reads 20000000 --reps 2 "$@" > $OUT/p$i.log 2>&1
  echo "pass $i ($SET) rc=$?"
done
python $R/scripts/pmc_summary.py $OUT
