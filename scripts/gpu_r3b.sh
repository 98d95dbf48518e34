# usage (GPU box): bash scripts/gpu_r3b.sh TAG -- the merged length-band launches: their tests, then bench.py's config 5 entry
TAG=${1:-r3b}
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_layouts.py tests/test_gpu_pairs.py -q -m gpu -x -k "bands or mixed or golden or cli or file" > gpurun_out/gputests_$TAG.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -30 gpurun_out/gputests_$TAG.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python - > gpurun_out/mixed_$TAG.json 2> gpurun_out/mixed_$TAG.err <<PY
import json, sys, os
sys.path.insert(0, 'kbbq-py_amd'); sys.path.insert(0, '.')
import torch, bench
from kbbq import _device as dev
for rowcost in (os.environ.get('ROWCOSTS') or '3').split(','):
    os.environ['KBBQ_K1_BAND_ROWCOST'] = rowcost
    r = bench.extra_mixed_lengths(torch, dev, 20_000_000, 10, 2)
    r.pop('layout', None); r['rowcost'] = rowcost
    print(json.dumps(r))
PY
rc=$?; echo "mixed rc=$rc"; tail -3 gpurun_out/mixed_$TAG.err; cat gpurun_out/mixed_$TAG.json
exit $rc
