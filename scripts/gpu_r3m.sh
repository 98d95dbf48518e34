# usage (GPU box): bash scripts/gpu_r3m.sh -- K1 with the context rows interleaved into the cycle rows (KBBQ_K1_IL=1): parity tests, then three alternating A/B rounds of the headline
mkdir -p gpurun_out
KBBQ_K1_IL=1 timeout -k 10 900 python -m pytest tests/test_gpu_layouts.py tests/test_gpu_pairs.py tests/test_gpu_parity.py -q -m gpu -x -k "not bench and not short_lived and not without_torch" > gpurun_out/gputests_r3m.log 2>&1
rc=$?; echo "pytest (KBBQ_K1_IL=1) rc=$rc"; tail -4 gpurun_out/gputests_r3m.log
[ $rc -eq 0 ] || exit $rc
for ROUND in 1 2 3; do
for IL in 0 1; do
KBBQ_K1_IL=$IL timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-extra --cpu-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('round $ROUND interleaved $IL: %.0f G  %.3f ms  K1 %.3f  K2 %.3f  verified %s' % (d['value']/1e9, d['ms_per_step'], d['kernels']['k1_accumulate']['avg_ms'], d['kernels']['k2_apply']['avg_ms'], d['verified']))"
done
done
