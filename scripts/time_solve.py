#!/usr/bin/env python3
"""Where the non-kernel part of a bench step goes: the host half of the model solve (K3) piece by piece."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'kbbq-py_amd'))
import numpy as np, torch
from kbbq import _device as dev, _solve, _native as N
n = 5_000_000
b = dev.ReadBatch.synthetic(0, n, n, seed=1)
t = dev.Tables(1, 300)
dev.accumulate(b, t)
torch.cuda.synchronize()
def T(f, reps=20):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): r = f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
host = t.to_host()
vec = _solve.vectors_from_tables(*host)
meanq, rg_e, rg_t, q_e, q_t, p_e, p_t, d_e, d_t = vec
print('tables.to_host (D2H 0.22 MB + 4 copies)   %.3f ms' % T(lambda: t.to_host()))
print('  raw buf.cpu()                           %.3f ms' % T(lambda: t.buf.cpu()))
print('vectors_from_tables (marginals, meanq)    %.3f ms' % T(lambda: _solve.vectors_from_tables(*host)))
print('combiln x4 (scipy gammaln, %d cells)   %.3f ms' % (p_e.size + d_e.size + q_e.size + rg_e.size, T(lambda: [_solve.combiln(a, c) for a, c in ((rg_e, rg_t), (q_e, q_t), (p_e, p_t), (d_e, d_t))])))
aux = np.concatenate([_solve.combiln(a, c).ravel() for a, c in ((rg_e, rg_t), (q_e, q_t), (p_e, p_t), (d_e, d_t))])
print('H2D aux (%.2f MB) + meanq                 %.3f ms' % (aux.nbytes / 1e6, T(lambda: (torch.from_numpy(aux).cuda(), torch.from_numpy(np.ascontiguousarray(meanq, dtype=np.int32)).cuda()))))
print('whole dev.solve                           %.3f ms' % T(lambda: dev.solve(t)))
print('tables.buf.zero_()                        %.3f ms' % T(lambda: t.buf.zero_()))
lut, shape, _, _ = dev.solve(t)
out = torch.empty_like(b.qual)
small = dev.ReadBatch.synthetic(0, 64, 64, seed=1); sout = torch.empty_like(small.qual)
print('accumulate wrapper on 64 reads (launch)   %.3f ms' % T(lambda: dev.accumulate(small, t, check=False)))
print('apply wrapper on 64 reads (launch)        %.3f ms' % T(lambda: dev.apply(small, lut, shape, out=sout, check=False)))
# finer: the pieces of dev.solve after the host vectors exist
lib = N.load(); R, S2, NQ = 1, 300, 43
d_aux = torch.from_numpy(aux).cuda(); d_meanq = torch.from_numpy(np.ascontiguousarray(meanq, dtype=np.int32)).cuda()
post_q = torch.empty(R * NQ, dtype=torch.int32, device='cuda')
lutb = torch.zeros(lib.kbbq_lut_bytes(R, NQ, S2), dtype=torch.uint8, device='cuda')
ctx = dev.context()
consts = dev._model_consts()
print('torch.zeros(lut) + empty(post_q)          %.3f ms' % T(lambda: (torch.zeros(lib.kbbq_lut_bytes(R, NQ, S2), dtype=torch.uint8, device='cuda'), torch.empty(R * NQ, dtype=torch.int32, device='cuda'))))
print('kbbq_solve_dev (3 kernels, async launch)  %.3f ms' % T(lambda: lib.kbbq_solve_dev(ctx.handle, N.ptr(t.buf), R, S2, 6, N.ptr(d_meanq), N.ptr(d_aux), N.ptr(consts), N.ptr(post_q), N.ptr(lutb), None)))
print('dev.context()                             %.3f ms' % T(lambda: dev.context()))
print('np.concatenate of 4 combiln results       %.3f ms' % T(lambda: np.concatenate([_solve.combiln(a, c).ravel() for a, c in ((rg_e, rg_t), (q_e, q_t), (p_e, p_t), (d_e, d_t))])))
