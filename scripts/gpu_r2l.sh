# usage (GPU box): bash scripts/gpu_r2l.sh -- the bench contract tests, then the mixed-length extras entry at full size
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bench" > gpurun_out/gputests_r2l.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/gputests_r2l.log
timeout -k 10 600 python - <<'PY'
import json, sys, os
sys.path.insert(0, os.getcwd())
import bench, torch
sys.path.insert(0, os.path.join(os.getcwd(), 'kbbq-py_amd'))
from kbbq import _device as dev
dev.warm_up()
for rep in range(2):
    r = bench.extra_mixed_lengths(torch, dev, 20_000_000, 5, 1)
    print(json.dumps(r, indent=1), flush=True)
    torch.cuda.empty_cache()
PY
