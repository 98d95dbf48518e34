# usage (GPU box): bash scripts/gpu_r3d.sh TAG -- K1 on narrow rows with two workgroups per CU: tests, config 5 A/B (KBBQ_K1_TWO x KBBQ_K1_BANDS), timeline
TAG=${1:-r3d}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_layouts.py tests/test_gpu_pairs.py tests/test_gpu_parity.py -q -m gpu -x -k "not full_size and not bench and not short_lived" > gpurun_out/gputests_$TAG.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -15 gpurun_out/gputests_$TAG.log
[ $rc -eq 0 ] || exit $rc
for TWO in 1 0 1 0; do
KBBQ_K1_TWO=$TWO timeout -k 10 300 python - >> gpurun_out/mixed_$TAG.json 2>> gpurun_out/mixed_$TAG.err <<PY
import json, sys, os
sys.path.insert(0, 'kbbq-py_amd'); sys.path.insert(0, '.')
import torch, bench
from kbbq import _device as dev
r = bench.extra_mixed_lengths(torch, dev, 20_000_000, 10, 2)
r.pop('layout', None); r['two'] = os.environ['KBBQ_K1_TWO']
print(json.dumps(r))
PY
rc=$?; echo "mixed (two=$TWO) rc=$rc"
[ $rc -eq 0 ] || exit $rc
done
python - <<PY
import json
for ln in open('gpurun_out/mixed_$TAG.json'):
    d = json.loads(ln)
    print('two', d['two'], 'verified', d['verified'], 'merged: %.0f G, %.3f ms, K1 %.3f (%.3f) K2 %.3f (%.3f)' % (d['value'] / 1e9, d['ms_per_step'], d['k1_accumulate_all_bands']['avg_ms'], d['k1_accumulate_all_bands']['frac'], d['k2_apply_all_bands']['avg_ms'], d['k2_apply_all_bands']['frac']),
          '| per band: %.0f G, %.3f ms, K1 %.3f K2 %.3f' % (d['launch_per_band']['value'] / 1e9, d['launch_per_band']['ms_per_step'], d['launch_per_band']['k1_accumulate_all_bands']['avg_ms'], d['launch_per_band']['k2_apply_all_bands']['avg_ms']))
PY
timeout -k 10 300 bash scripts/e2e_timeline.sh 8000000 > gpurun_out/timeline_$TAG.log 2>&1
rc=$?; echo "timeline rc=$rc"; cat gpurun_out/timeline_$TAG.log
exit $rc
