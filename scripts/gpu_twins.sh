# usage (GPU box): bash scripts/gpu_twins.sh -- single-end reads two to a row: the kernel fuzz campaign (a quarter of its cases single-end), then timings
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 400 python tests/tools/fuzz_gpu.py --seconds 150 > gpurun_out/fuzz_twins.log 2>&1; echo "fuzz rc=$?"; tail -2 gpurun_out/fuzz_twins.log
timeout -k 10 600 python - <<'PY'
import json, sys, os
sys.path.insert(0, os.getcwd())
import bench, torch
sys.path.insert(0, os.path.join(os.getcwd(), 'kbbq-py_amd'))
from kbbq import _device as dev, parallel
dev.warm_up()
for rep in range(2):
    for se in (True, False):
        r = bench.extra_layout(torch, dev, parallel, 50_000_000, 10, 2, 'packed', single_end=se)
        print(se, r['layout'], round(r['value'] / 1e9, 1), round(r['ms_per_step'], 3), round(r['k1_accumulate']['avg_ms'], 3), round(r['k2_apply']['avg_ms'], 3), flush=True)
        torch.cuda.empty_cache()
PY
