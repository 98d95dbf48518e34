# usage: bash scripts/gpu_round.sh [tag]   (run on the GPU box via gpurun)
TAG=${1:-r}
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -q -m gpu > gpurun_out/gputests_$TAG.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/gputests_$TAG.log
timeout -k 10 600 python bench.py --steps 5 --warmup 2 --cpu-sample ${CPU_SAMPLE:-0} > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; echo "bench rc=$?"; cat gpurun_out/bench_$TAG.json; tail -3 gpurun_out/bench_$TAG.err
