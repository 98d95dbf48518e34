# usage (GPU box): bash scripts/gpu_k2tile.sh -- K2 with short-lived workgroups against the persistent kernel (KBBQ_K2_TILE=0): layout tests, kernel timings at 1 / 8 read groups, bench line both ways
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_layouts.py tests/test_gpu_pairs.py -x -q -m gpu > gpurun_out/gputests_k2t.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/gputests_k2t.log
for RG in 1 8; do
echo "--- rgs $RG: default (short-lived K2) / KBBQ_K2_TILE=0"
timeout -k 10 200 python scripts/time_kernels.py --reads 50000000 --packed --rgs $RG 2>&1 | tail -1
KBBQ_K2_TILE=0 timeout -k 10 200 python scripts/time_kernels.py --reads 50000000 --packed --rgs $RG 2>&1 | tail -1
done
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --cpu-sample 0 --no-extra > gpurun_out/bench_k2t.json 2> gpurun_out/bench_k2t.err; echo "bench rc=$?"; python -c "import json; d=json.load(open('gpurun_out/bench_k2t.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['kernels']['k1_accumulate']['avg_ms'], d['kernels']['k2_apply']['avg_ms'])"
KBBQ_K2_TILE=0 timeout -k 10 300 python bench.py --steps 10 --warmup 3 --cpu-sample 0 --no-extra > gpurun_out/bench_k2t0.json 2>> gpurun_out/bench_k2t.err; echo "bench (persistent K2) rc=$?"; python -c "import json; d=json.load(open('gpurun_out/bench_k2t0.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['kernels']['k1_accumulate']['avg_ms'], d['kernels']['k2_apply']['avg_ms'])"
