# usage (GPU box): bash scripts/gpu_fuzz_all.sh SECONDS -- the three randomised campaigns against the oracle, one after the other
R=$GRAFT_REPO_ROOT
SEC=${1:-300}
cd $R
for f in fuzz_gpu fuzz_gpu_aligned fuzz_gpu_cli; do
  timeout -k 10 $((SEC + 240)) python tests/tools/$f.py --seconds $SEC > gpurun_out/$f.log 2>&1; rc=$?
  echo "$f rc=$rc: $(tail -1 gpurun_out/$f.log)"
  if [ $rc -ne 0 ]; then tail -25 gpurun_out/$f.log; exit $rc; fi
done
