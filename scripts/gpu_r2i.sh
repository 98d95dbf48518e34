# usage (GPU box): bash scripts/gpu_r2i.sh -- file-path tests, then the bench's file-path and aligned-kernel extras at two sizes
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_pairs.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/gputests_r2i.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/gputests_r2i.log
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
for i in range(2):
    r = bench.extra_file_path(torch, dev, n=8_000_000)
    print(json.dumps(r), flush=True)
PY
