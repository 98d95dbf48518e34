# usage (GPU box): bash scripts/gpu_r2c.sh TAG -- the solve / layout / bench tests, then the bench line
TAG=${1:-r2c}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_layouts.py -x -q -m gpu > gpurun_out/gputests_$TAG.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/gputests_$TAG.log
timeout -k 10 400 python bench.py --steps 10 --warmup 3 --cpu-sample 0 --no-extra > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; echo "bench rc=$?"; cat gpurun_out/bench_$TAG.json; tail -5 gpurun_out/bench_$TAG.err
KBBQ_HOST_SOLVE=1 timeout -k 10 400 python bench.py --steps 10 --warmup 3 --cpu-sample 0 --no-extra > gpurun_out/bench_${TAG}_hostsolve.json 2>> gpurun_out/bench_$TAG.err; echo "bench (host solve) rc=$?"; cat gpurun_out/bench_${TAG}_hostsolve.json
