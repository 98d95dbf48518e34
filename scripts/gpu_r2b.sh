# usage (GPU box): bash scripts/gpu_r2b.sh TAG -- whole GPU suite, then the default bench line with its extra object
TAG=${1:-r2b}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/gputests_$TAG.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/gputests_$TAG.log
timeout -k 10 400 python bench.py --steps 5 --warmup 2 > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; echo "bench rc=$?"; cat gpurun_out/bench_$TAG.json; tail -5 gpurun_out/bench_$TAG.err
