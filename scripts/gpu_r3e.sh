# usage (GPU box): bash scripts/gpu_r3e.sh TAG -- K6 fused into K1 + the sharded `kbbq benchmark -f`: tests, the aligned-read campaign, bench entries
TAG=${1:-r3e}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_bqsr.py tests/test_gpu_benchmark.py tests/test_gpu_pairs.py -q -m gpu -x -k "not bench_launches" > gpurun_out/gputests_$TAG.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -25 gpurun_out/gputests_$TAG.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tests/tools/fuzz_gpu_aligned.py --seconds 150 --seed 31 > gpurun_out/fuzz_aligned_$TAG.log 2>&1
rc=$?; echo "fuzz rc=$rc"; tail -3 gpurun_out/fuzz_aligned_$TAG.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python - > gpurun_out/aligned_$TAG.json 2> gpurun_out/aligned_$TAG.err <<PY
import json, sys
sys.path.insert(0, 'kbbq-py_amd'); sys.path.insert(0, '.')
import torch, bench
from kbbq import _device as dev
r = bench.extra_aligned(torch, dev, n=16_000_000)
r.pop('bytes_per_base', None)
print(json.dumps(r))
PY
rc=$?; echo "aligned rc=$rc"; tail -3 gpurun_out/aligned_$TAG.err
python - <<PY
import json
d = json.loads(open('gpurun_out/aligned_$TAG.json').read())
for k, v in d.items():
    print(k, json.dumps(v)[:300])
PY
exit $rc
