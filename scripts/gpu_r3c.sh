# usage (GPU box): bash scripts/gpu_r3c.sh TAG -- merged length-band launches + the torch-free command line: tests, config 5, start-up timeline
TAG=${1:-r3c}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_layouts.py tests/test_gpu_pairs.py tests/test_gpu_parity.py -q -m gpu -x -k "bands or mixed or golden or without_torch or known_answers or error_behaviour or single_end or checked" > gpurun_out/gputests_$TAG.log 2>&1
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
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 bash scripts/e2e_timeline.sh 8000000 > gpurun_out/timeline_$TAG.log 2>&1
rc=$?; echo "timeline rc=$rc"; cat gpurun_out/timeline_$TAG.log
exit $rc
