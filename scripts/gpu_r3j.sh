# usage (GPU box): bash scripts/gpu_r3j.sh -- the fused aligned-read tally with 1 / 2 / 4 trash rows, alternating processes
for NT in 4 1 2 4 1 2; do
KBBQ_K1_NTRASH=$NT timeout -k 10 200 python - <<PY
import sys, os
sys.path.insert(0, 'kbbq-py_amd'); sys.path.insert(0, '.')
import torch, bench
from kbbq import _device as dev
r = bench.extra_aligned(torch, dev, n=16_000_000)
print('trash rows', os.environ['KBBQ_K1_NTRASH'], 'fused %.3f ms verified %s | K4 tally %.3f  K6 %.3f  K1 on canonical %.3f' % (r['k61_fused_tally']['avg_ms'], r['k61_fused_tally']['verified'], r['k4_find_errors_tally']['avg_ms'], r['k6_canonical_reads']['avg_ms'], r['k1_on_canonical_reads']['avg_ms']), flush=True)
PY
done
