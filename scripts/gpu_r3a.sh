# usage (GPU box): bash scripts/gpu_r3a.sh TAG -- GPU tests (stop at the first failure), then bench.py as the driver runs it
TAG=${1:-r3a}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -q -m gpu -x > gpurun_out/gputests_$TAG.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -40 gpurun_out/gputests_$TAG.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 700 python bench.py --steps 10 --warmup 3 > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
rc=$?; echo "bench rc=$rc"; tail -5 gpurun_out/bench_$TAG.err; python - <<PY
import json
d = json.loads(open('gpurun_out/bench_$TAG.json').read().strip().splitlines()[-1])
x = d.pop('extra', {})
print(json.dumps(d)[:3000])
for k, v in x.items():
    print(k, json.dumps(v)[:1200])
PY
exit $rc
