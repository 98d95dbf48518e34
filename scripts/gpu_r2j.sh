# usage (GPU box): bash scripts/gpu_r2j.sh -- aligned-read tests, then the bench's aligned-kernel extras
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_bqsr.py tests/test_gpu_benchmark.py tests/test_gpu_report.py -x -q -m gpu > gpurun_out/gputests_r2j.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/gputests_r2j.log
timeout -k 10 600 python - <<'PY'
import json, sys, os
sys.path.insert(0, os.getcwd())
import bench, torch
sys.path.insert(0, os.path.join(os.getcwd(), 'kbbq-py_amd'))
from kbbq import _device as dev
dev.warm_up()
for n in (4_000_000, 16_000_000):
    r = bench.extra_aligned(torch, dev, n=n, G=200_000_000)
    print(n, {k: (round(v['avg_ms'], 3), round(v['frac'], 3)) for k, v in r.items() if isinstance(v, dict)}, flush=True)
    torch.cuda.empty_cache()
PY
timeout -k 10 300 python tests/tools/fuzz_gpu_aligned.py --seconds 120 > gpurun_out/fuzz_aligned_r2j.log 2>&1; echo "fuzz rc=$?"; tail -3 gpurun_out/fuzz_aligned_r2j.log
