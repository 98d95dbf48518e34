#!/usr/bin/env python3
"""dev.solve piece by piece at several read-group counts (the solve grows with R x 43 x (2S + 16) cells)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'kbbq-py_amd'))
import numpy as np, torch
from kbbq import _device as dev, _solve, _native as N
def T(f, reps=20):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for R in (1, 8, 32):
    n = 4_000_000
    b = dev.ReadBatch.synthetic(0, n, n, seed=1, nrg=R)
    t = dev.Tables(R, 300); dev.accumulate(b, t); torch.cuda.synchronize()
    host = t.to_host(); vec = _solve.vectors_from_tables(*host)
    meanq, rg_e, rg_t, q_e, q_t, p_e, p_t, d_e, d_t = vec
    E = np.concatenate([x.ravel() for x in (rg_e, q_e, p_e, d_e)]); Tt = np.concatenate([x.ravel() for x in (rg_t, q_t, p_t, d_t)])
    print('R=%2d: to_host %.3f  vectors %.3f  concat %.3f  combiln(%d cells) %.3f  whole dev.solve %.3f ms' % (
        R, T(lambda: t.to_host()), T(lambda: _solve.vectors_from_tables(*host)),
        T(lambda: (np.concatenate([x.ravel() for x in (rg_e, q_e, p_e, d_e)]), np.concatenate([x.ravel() for x in (rg_t, q_t, p_t, d_t)]))),
        E.size, T(lambda: _solve.combiln(E, Tt)), T(lambda: dev.solve(t))), flush=True)
    lib = N.load(); NQ = 43; S2 = 300
    aux = _solve.combiln(E, Tt)
    hostb = np.concatenate([aux.view(np.uint8), np.ascontiguousarray(meanq, dtype=np.int32).view(np.uint8)])
    d_host = torch.from_numpy(hostb).cuda(); d_aux, d_meanq = d_host[:aux.size * 8], d_host[aux.size * 8:]
    post_q = torch.empty(R * NQ, dtype=torch.int32, device='cuda'); lutb = torch.zeros(lib.kbbq_lut_bytes(R, NQ, S2), dtype=torch.uint8, device='cuda')
    ctx = dev.context(); consts = dev._model_consts()
    print('      H2D %.3f  zeros(lut %d KB) %.3f  K3 kernels (launch + run, synchronous) %.3f ms' % (
        T(lambda: torch.from_numpy(hostb).cuda()), lutb.numel() // 1024, T(lambda: torch.zeros(lib.kbbq_lut_bytes(R, NQ, S2), dtype=torch.uint8, device='cuda')),
        T(lambda: (lib.kbbq_solve_dev(ctx.handle, N.ptr(t.buf), R, S2, 6, N.ptr(d_meanq), N.ptr(d_aux), N.ptr(consts), N.ptr(post_q), N.ptr(lutb), None), torch.cuda.synchronize()))), flush=True)
