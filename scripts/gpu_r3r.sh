# usage (GPU box): bash scripts/gpu_r3r.sh -- K2's tiles by XCD (KBBQ_K2_XCD_TILES): layout / pair / parity tests on the new default, then A/B on the headline
# layout, on 8 read groups and on config 5's bands
mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests/test_gpu_layouts.py tests/test_gpu_pairs.py tests/test_gpu_parity.py -q -m gpu -x > gpurun_out/gputests_r3r.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/gputests_r3r.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python scripts/time_xcd_streams.py 2>&1 | tee gpurun_out/xcd_streams.txt
for x in 0 1 0 1; do
KBBQ_K2_XCD_TILES=$x timeout -k 10 300 python - <<'PY'
import os, sys
sys.path.insert(0, 'kbbq-py_amd'); sys.path.insert(0, '.')
import torch, bench
from kbbq import _device as dev, parallel
c3 = bench.extra_config3(torch, dev, parallel, 50_000_000, 10, 2)
c5 = bench.extra_mixed_lengths(torch, dev, 20_000_000, 10, 2)
print('KBBQ_K2_XCD_TILES=%s  8 read groups: K2 %.3f ms (verified %s)   config 5: K2 %.3f ms K1 %.3f ms, %.1f Gbases/s (verified %s)' % (
    os.environ['KBBQ_K2_XCD_TILES'], c3['k2_apply']['avg_ms'], c3['verified'], c5['k2_apply_all_bands']['avg_ms'], c5['k1_accumulate_all_bands']['avg_ms'], c5['value'] / 1e9, c5['verified']), flush=True)
PY
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/xcd_tiles_c3c5.txt
