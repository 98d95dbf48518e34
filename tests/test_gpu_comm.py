"""
GPU test (-m gpu) of the C ABI's exchange step (include/kbbq_hip.h "the exchange step of the sharded path without
torch"): kbbq_comm_unique_id / kbbq_comm_create / kbbq_allreduce_tables / kbbq_comm_destroy over RCCL.  A one-GPU box
can only form a single-rank communicator (RCCL refuses two ranks on one device); the N-rank form is the same call
sequence and is what the driver's multi-GPU node exercises through bench.py's RCCL group.
"""
import ctypes

import numpy as np
import pytest

from test_gpu_parity import dev                      # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


def test_single_rank_communicator_sums_the_tables_in_place(dev):
    import torch
    from kbbq import _native as N
    lib = N.load()
    ctx = dev.context()
    ident = ctypes.create_string_buffer(128)
    N.check(lib.kbbq_comm_unique_id(ident))
    assert any(ident.raw)
    comm = ctypes.c_void_p()
    N.check(lib.kbbq_comm_create(ctx.handle, ident, 1, 0, ctypes.byref(comm)))
    try:
        b = dev.ReadBatch.synthetic(0, 4000, 4000, seed=2, nrg=2)
        t = dev.Tables(2, 300)
        dev.accumulate(b, t)
        before = t.buf.clone()
        N.check(lib.kbbq_allreduce_tables(comm, N.ptr(t.buf), t.buf.numel()))      # on the context's stream, after K1
        lut, shape = dev.solve_lut(t)                                              # ... and before the solve
        torch.cuda.synchronize()
        assert torch.equal(t.buf, before) and int(before.sum()) > 0
        N.check(lib.kbbq_allreduce_tables(comm, None, 0))
        with pytest.raises(ValueError):
            N.check(lib.kbbq_allreduce_tables(None, N.ptr(t.buf), 1))
    finally:
        N.check(lib.kbbq_comm_destroy(comm))
    bad = ctypes.c_void_p()
    with pytest.raises(ValueError):
        N.check(lib.kbbq_comm_create(ctx.handle, ident, 2, 2, ctypes.byref(bad)))   # rank outside 0..nranks-1
