# usage (GPU box): bash scripts/gpu_r3l.sh -- the headline's K1 (2 x 150 bp, chunk-position-major) with 1 / 4 / 8 trash rows, alternating processes
for NT in 1 8 4 1 8 4; do
KBBQ_K1_NTRASH_KM=$NT timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-extra --cpu-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('trash rows (KM) $NT: %.0f G  %.3f ms  K1 %.3f  K2 %.3f  verified %s' % (d['value']/1e9, d['ms_per_step'], d['kernels']['k1_accumulate']['avg_ms'], d['kernels']['k2_apply']['avg_ms'], d['verified']))"
done
