# usage (GPU box): bash scripts/gpu_r3i.sh TAG -- copies of K1's cycle table for narrow rows: tests, then config 5 and short-read batches A/B (KBBQ_K1_POSCOPIES)
TAG=${1:-r3i}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_layouts.py tests/test_gpu_pairs.py tests/test_gpu_parity.py tests/test_gpu_bqsr.py -q -m gpu -x -k "not full_size and not bench and not headline and not short_lived" > gpurun_out/gputests_$TAG.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -15 gpurun_out/gputests_$TAG.log
[ $rc -eq 0 ] || exit $rc
for PC in 4:8 1:1 4:1 1:8 4:8 1:1; do
KBBQ_K1_POSCOPIES=${PC%%:*} KBBQ_K1_NTRASH=${PC##*:} timeout -k 10 300 python - <<PY
import json, sys, os, time
sys.path.insert(0, 'kbbq-py_amd'); sys.path.insert(0, '.')
import torch, bench
from kbbq import _device as dev
d = bench.extra_mixed_lengths(torch, dev, 20_000_000, 10, 2)
print('copies', os.environ['KBBQ_K1_POSCOPIES'], 'trash rows', os.environ['KBBQ_K1_NTRASH'], 'config5 verified', d['verified'], 'merged: %.0f G, %.3f ms, K1 %.3f (%.3f) K2 %.3f' % (d['value'] / 1e9, d['ms_per_step'], d['k1_accumulate_all_bands']['avg_ms'], d['k1_accumulate_all_bands']['frac'], d['k2_apply_all_bands']['avg_ms']),
      '| per band K1 %.3f' % d['launch_per_band']['k1_accumulate_all_bands']['avg_ms'], flush=True)
# short reads of one length, one read per row on 4-bit planes (and 2 x 50 mate-pair rows)
ctx = dev.context()
for L, n in ((36, 40_000_000), (50, 40_000_000), (75, 30_000_000), (100, 20_000_000), (150, 20_000_000)):
    b = dev.ReadBatch.synthetic(0, n, n, seed=1, len_lo=L, len_hi=L)
    for pairs in (False, True):
        laid = dev.lay_out(b, 1, L, packed=True, pairs=pairs)
        t = dev.Tables(1, 2 * L)
        dev.accumulate(laid, t)
        ctx.kernel_ms(0, reset=True); ctx.timing(True)
        for _ in range(5):
            dev.accumulate(laid, t, check=False)
        ctx.timing(False)
        ms, k = ctx.kernel_ms(0)
        print('   %3d bp x %d M %-10s pitch %3d: K1 %.3f ms = %.0f Gbases/s' % (L, n // 1000000, 'pair rows' if pairs else 'rows', laid.pitch, ms / k, n * L / (ms / k) / 1e6), flush=True)
        del laid
    del b
    torch.cuda.empty_cache()
PY
rc=$?; [ $rc -eq 0 ] || exit $rc
done
