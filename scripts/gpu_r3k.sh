# usage (GPU box): bash scripts/gpu_r3k.sh TAG -- after the last kernel edits: aligned-read tests, the fused tally's time, PMC traffic passes, bench line
TAG=${1:-r3k}
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_bqsr.py tests/test_gpu_benchmark.py -q -m gpu -x > gpurun_out/gputests_$TAG.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/gputests_$TAG.log
[ $rc -eq 0 ] || exit $rc
bash scripts/gpu_r3j.sh 2>&1 | grep "trash rows" | head -3
