# usage (GPU box): bash scripts/gpu_r3_check.sh -- what the driver runs at round end: GPU tests, smoke(), bench.py
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -q -m gpu -x > gpurun_out/gputests_check.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/gputests_check.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()"
rc=$?; echo "smoke rc=$rc"
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py > gpurun_out/bench_check.json 2> gpurun_out/bench_check.err
rc=$?; echo "bench rc=$rc"; python - <<PY
import json
d = json.loads(open('gpurun_out/bench_check.json').read().strip().splitlines()[-1])
x = d.pop('extra')
print('value %.4g  ms %.3f  verified %s  frac %.3f  traffic %s' % (d['value'], d['ms_per_step'], d['verified'], d['roofline']['frac'], d['roofline']['traffic']))
print({k: (v.get('value'), v.get('verified'), v.get('error')) for k, v in x.items()})
PY
exit $rc
