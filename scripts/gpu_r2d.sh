# usage (GPU box): bash scripts/gpu_r2d.sh TAG [fuzz seconds] -- aligned-read rows: tests, a randomised campaign, then K4 / K5 / K6 timings (new form, and KBBQ_K4=v1)
TAG=${1:-r2d}
FUZZ=${2:-120}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_benchmark.py tests/test_gpu_bqsr.py -x -q -m gpu > gpurun_out/gputests_$TAG.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/gputests_$TAG.log
timeout -k 10 $((FUZZ + 120)) python tests/tools/fuzz_gpu_aligned.py --seconds $FUZZ --seed 11 > gpurun_out/fuzz_aligned_$TAG.log 2>&1; echo "fuzz rc=$?"; tail -3 gpurun_out/fuzz_aligned_$TAG.log
for INS in 0.05 0.0 0.2; do
  echo "--- ins $INS"
  timeout -k 10 200 python scripts/time_benchmark_path.py --ins $INS 2>&1 | grep -v "^K1\|errors flagged"
  KBBQ_K4=v1 timeout -k 10 200 python scripts/time_benchmark_path.py --ins $INS 2>&1 | grep "K4" | grep -v flags
done
