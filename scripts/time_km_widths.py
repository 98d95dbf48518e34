#!/usr/bin/env python3
"""K1 on mate-pair rows of 2 x 50 / 75 / 100 / 125 / 150 bp (4-bit planes): the chunk-position-major cycle table (KJ = 7 / 10 / 13 / 16 / 19)
against the position-major form (KBBQ_K1_KM=19 keeps only the 19-chunk form of round 2), same process, alternating; tables compared.
usage (GPU box): python scripts/time_km_widths.py [Mbases per batch]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'kbbq-py_amd'))
import torch
from kbbq import _device as dev
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
ctx = dev.context()
for S in (50, 75, 100, 125, 150):
    n = (mb * 1_000_000 // S) & ~1
    b = dev.ReadBatch.synthetic(0, n, n, seed=1, len_lo=S, len_hi=S)
    rows = dev.lay_out(b, 1, S, packed=True)
    del b
    torch.cuda.empty_cache()
    res = {}
    for rep in range(3):
        for mode, env in (('position-major', '19' if S != 150 else '0'), ('chunk-position-major', None)):
            for nt in ((None,) if mode == 'position-major' else (None, '1')):
                if env is None: os.environ.pop('KBBQ_K1_KM', None)
                else: os.environ['KBBQ_K1_KM'] = env
                if nt is None: os.environ.pop('KBBQ_K1_KM_NTRASH', None)
                else: os.environ['KBBQ_K1_KM_NTRASH'] = nt
                t = dev.Tables(1, 2 * S)
                dev.accumulate(rows, t)
                ctx.kernel_ms(0, reset=True); ctx.timing(True)
                for _ in range(5):
                    dev.accumulate(rows, t, check=False)
                ctx.timing(False)
                ms = ctx.kernel_ms(0)[0] / 5
                key = mode + ('' if nt is None else ', 1 trash row')
                res.setdefault(key, []).append(ms)
                res.setdefault('tables ' + key, t.buf.clone())
    same = all(torch.equal(res['tables position-major'], v) for k, v in res.items() if k.startswith('tables'))
    print('2 x %3d bp, %d M reads, %s: ' % (S, n // 1_000_000, rows.describe()) +
          '   '.join('%s %s ms = %.0f Gbases/s' % (k, '/'.join('%.3f' % x for x in v), n * S / min(v) / 1e6) for k, v in res.items() if not k.startswith('tables')) +
          '   same tables %s' % same, flush=True)
    del rows, res
    torch.cuda.empty_cache()
