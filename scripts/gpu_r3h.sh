# usage (GPU box): ROWCOSTS=2,3,4 bash scripts/gpu_r3h.sh TAG -- config 5's merged K1: the per-row term of the bands' shares of the workgroups
TAG=${1:-r3h}
mkdir -p gpurun_out
timeout -k 10 600 python - > gpurun_out/mixed_$TAG.txt 2> gpurun_out/mixed_$TAG.err <<PY
import json, sys, os
sys.path.insert(0, 'kbbq-py_amd'); sys.path.insert(0, '.')
import torch, bench
from kbbq import _device as dev
for rep in range(2):
    for rowcost in (os.environ.get('ROWCOSTS') or '3').split(','):
        os.environ['KBBQ_K1_BAND_ROWCOST'] = rowcost
        d = bench.extra_mixed_lengths(torch, dev, 20_000_000, 10, 2)
        print('rowcost', rowcost, 'verified', d['verified'], 'merged: %.0f G, %.3f ms, K1 %.3f (%.3f) K2 %.3f (%.3f) outside %.3f' % (d['value'] / 1e9, d['ms_per_step'], d['k1_accumulate_all_bands']['avg_ms'], d['k1_accumulate_all_bands']['frac'], d['k2_apply_all_bands']['avg_ms'], d['k2_apply_all_bands']['frac'], d['outside_the_kernels_ms']), flush=True)
PY
rc=$?; cat gpurun_out/mixed_$TAG.txt; tail -2 gpurun_out/mixed_$TAG.err; exit $rc
