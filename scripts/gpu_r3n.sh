# usage (GPU box): bash scripts/gpu_r3n.sh -- the one-pass BAM-sourced tally (kbbq_tally_aligned_dev): its tests, then the aligned-read kernels of bench.py
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_bqsr.py tests/test_gpu_benchmark.py -q -m gpu -x > gpurun_out/gputests_r3n.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -15 gpurun_out/gputests_r3n.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python - <<'PY' 2>&1 | tee gpurun_out/aligned_r3n.txt
import sys, json
sys.path.insert(0, 'kbbq-py_amd'); sys.path.insert(0, '.')
import torch, bench
from kbbq import _device as dev
for rep in range(2):
    out = bench.extra_aligned(torch, dev, n=16_000_000, G=200_000_000)
    for k, v in out.items():
        if isinstance(v, dict) and 'avg_ms' in v:
            print('%-45s %.3f ms  frac %.3f %s' % (k, v['avg_ms'], v['frac'], v.get('verified', '')))
    print(json.dumps(out['whole_tally_ms']))
PY
