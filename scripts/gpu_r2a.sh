# usage (GPU box): bash scripts/gpu_r2a.sh TAG -- round-2 first pass: new layout tests, kernel timings of every layout, bench
TAG=${1:-r2a}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 500 python -m pytest tests/test_gpu_layouts.py -x -q -m gpu > gpurun_out/gputests_$TAG.log 2>&1; echo "layout tests rc=$?"; tail -15 gpurun_out/gputests_$TAG.log
for L in "" "--pairs" "--packed"; do
  timeout -k 10 200 python scripts/time_kernels.py --reads 20000000 $L 2>&1 | tail -1
done
timeout -k 10 200 python scripts/time_kernels.py --reads 20000000 --rgs 8 --packed 2>&1 | tail -1
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-extra > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; echo "bench rc=$?"; cat gpurun_out/bench_$TAG.json; tail -3 gpurun_out/bench_$TAG.err
