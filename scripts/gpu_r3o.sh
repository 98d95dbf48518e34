# usage (GPU box): bash scripts/gpu_r3o.sh -- config 5's merged K1 launch: when do the bands' workgroups end? (KBBQ_K1_BANDS_DBG), for several
# slopes of the per-chunk cost in the bands' shares (KBBQ_K1_BAND_SLOPE) and costs of a band on 8 context-table copies (KBBQ_K1_BAND_DN8)
mkdir -p gpurun_out
for cfg in "0 1.0" "0.01 1.0" "0.01 1.06" "0.01 1.1" "0.015 1.06" "0.01 1.06" "0 1.0"; do
set -- $cfg
KBBQ_K1_BAND_SLOPE=$1 KBBQ_K1_BAND_DN8=$2 timeout -k 10 300 python - > gpurun_out/bands_dbg_$1_$2.txt 2>&1 <<'PY'
import sys
sys.path.insert(0, 'kbbq-py_amd'); sys.path.insert(0, '.')
import torch, bench
from kbbq import _device as dev
out = bench.extra_mixed_lengths(torch, dev, 20_000_000, 10, 2)
print('verified %s value %.1f G  ms %.3f  K1 %.3f (%.3f)  K2 %.3f (%.3f)' % (out['verified'], out['value'] / 1e9, out['ms_per_step'], out['k1_accumulate_all_bands']['avg_ms'],
      out['k1_accumulate_all_bands']['frac'], out['k2_apply_all_bands']['avg_ms'], out['k2_apply_all_bands']['frac']))
PY
echo "slope $1 dn8 $2: $(tail -1 gpurun_out/bands_dbg_$1_$2.txt)"
done
KBBQ_K1_BANDS_DBG=1 timeout -k 10 300 python - 2>&1 <<'PY' | tail -9
import sys
sys.path.insert(0, 'kbbq-py_amd'); sys.path.insert(0, '.')
import torch, bench
from kbbq import _device as dev
out = bench.extra_mixed_lengths(torch, dev, 20_000_000, 2, 1)
PY
