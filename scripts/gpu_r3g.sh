# usage (GPU box): bash scripts/gpu_r3g.sh TAG -- K1's 4-bit by-product: tests, smoke, then the two bench entries it concerns
TAG=${1:-r3g}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_layouts.py tests/test_gpu_parity.py -q -m gpu -x -k "not full_size and not bench_launches and not headline" > gpurun_out/gputests_$TAG.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -25 gpurun_out/gputests_$TAG.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()"
rc=$?; echo "smoke rc=$rc"
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python - <<PY
import json, sys
sys.path.insert(0, 'kbbq-py_amd'); sys.path.insert(0, '.')
import torch, bench
from kbbq import _device as dev, parallel
for rep in range(2):
    for emit in (False, True):
        r = bench.extra_layout(torch, dev, parallel, 50_000_000, 10, 2, 'reads', emit=emit)
        print(emit, r['verified'], '%.0f G' % (r['value'] / 1e9), '%.3f ms' % r['ms_per_step'], 'K1 %.3f K2 %.3f' % (r['k1_accumulate']['avg_ms'], r['k2_apply']['avg_ms']), r['layout'])
PY
