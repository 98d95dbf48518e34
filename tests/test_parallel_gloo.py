"""
Multi-process CPU test of the N > 1 host logic (world_size 2 and 8, gloo): shard ranges,
first-appearance read-group merge, global max length and the sum-allreduce of the count
tables.  K1 itself cannot run here (no GPU): each rank's tables come from the CPU oracle
(tests may use it), which is exactly what K1 produces on the GPU box
(tests/test_gpu_parity.py).  After the allreduce every rank must hold the reference's
golden tables.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, load_golden


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, name, q):
    import sys
    for p in (os.path.join(ROOT, 'kbbq-py_amd'), os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
        sys.path.insert(0, p)
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import oracle as O
        from kbbq import parallel, _solve
        info, gold = load_golden(name)
        c = info['case']
        lo, hi = parallel.shard_range(c['n'], rank, world)
        assert lo % 2 == 0
        seq, cseq, qual, meta = O.synth(lo, hi - lo, c['n'], c['seed'], c['len_lo'], c['len_hi'],
                                        c['nrg'], c['qlo'], c['qhi'])
        # read-group names of this shard in LOCAL first-appearance order -> global ids
        rg_local = []
        for r in ((meta >> 16) & 0x7FFF).tolist():
            nm = 'g%d' % r
            if nm not in rg_local:
                rg_local.append(nm)
        names, remap = parallel.merge_rg_maps(rg_local)
        assert names == ['g%d' % i for i in range(c['nrg'])]
        local_id = {nm: i for i, nm in enumerate(rg_local)}
        rg_glob = np.array([remap[local_id['g%d' % r]] for r in ((meta >> 16) & 0x7FFF)], dtype=np.uint32)
        assert np.array_equal(rg_glob, (meta >> 16) & 0x7FFF)
        S = parallel.max_over_ranks(int((meta & 0xFFFF).max()))
        assert S == c['len_hi']
        R = len(names)
        v = O.accumulate(seq, cseq, qual, meta, R, S)        # stand-in for K1 on this rank's shard
        buf = torch.from_numpy(np.concatenate([v[5].ravel(), v[6].ravel(), v[7].ravel(), v[8].ravel()]))
        parallel.allreduce_tables(buf)
        h = buf.numpy()
        npos, ndn = R * 43 * 2 * S, R * 43 * 16
        tabs = (h[:npos].reshape(R, 43, 2 * S), h[npos:2 * npos].reshape(R, 43, 2 * S),
                h[2 * npos:2 * npos + ndn].reshape(R, 43, 16), h[2 * npos + ndn:].reshape(R, 43, 16))
        vec = _solve.vectors_from_tables(*tabs)
        keys = ['meanq', 'rg_errs', 'rg_total', 'q_errs', 'q_total', 'pos_errs', 'pos_total',
                'dinuc_errs', 'dinuc_total']
        ok = all(np.array_equal(a, gold[k]) for a, k in zip(vec, keys))
        q.put((rank, ok, hi - lo))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('name,world', [('c1_10k_1rg', 2), ('c3cut_2k_8rg', 2), ('c3cut_2k_8rg', 8), ('c5cut_2k_mixed', 8)])
def test_sharded_tables_allreduce_to_reference(name, world):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    info, _ = load_golden(name)
    assert sorted(r[0] for r in res) == list(range(world)) and all(r[1] for r in res)
    assert sum(r[2] for r in res) == info['case']['n']


def test_shard_range():
    from kbbq import parallel
    for n in (0, 1, 2, 7, 10, 1000, 1001, 50_000_000):
        for w in (1, 2, 3, 4, 8):
            spans = [parallel.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c and a <= b
            assert all(lo % 2 == 0 or lo == n for lo, _ in spans)


def _agree_worker(rank, world, port, q):
    import sys
    for p in (os.path.join(ROOT, 'kbbq-py_amd'), os.path.join(ROOT, 'tests')):
        sys.path.insert(0, p)
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from kbbq import parallel
        out = []
        parallel.raise_first_error(None)                                  # nobody has an error: returns
        try:                                                              # rank 1 has one: everybody raises it
            parallel.raise_first_error(IndexError('read 70') if rank == 1 else None, 70)
        except IndexError as e:
            out.append(('index', str(e)))
        try:                                                              # both have one: the smaller read index wins
            exc = TypeError('read 5') if rank == 1 else IndexError('read 9')
            parallel.raise_first_error(exc, 5 if rank == 1 else 9)
        except (TypeError, IndexError) as e:
            out.append((type(e).__name__, str(e)))
        order = []
        parallel.in_rank_order(lambda: order.append(rank))
        out.append(('world', parallel.world_rank()))
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_error_agreement_and_rank_order():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_agree_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in (0, 1):
        assert res[rank][0] == ('index', 'read 70')
        assert res[rank][1] == ('TypeError', 'read 5')
        assert res[rank][2] == ('world', (2, rank))


def _pack_worker(rank, world, port, fa, fb, q):
    import sys
    for p in (os.path.join(ROOT, 'kbbq-py_amd'), os.path.join(ROOT, 'oracle')):
        sys.path.insert(0, p)
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from kbbq import fastx, parallel
        p = fastx.pack_pair(fa, fb, True, shard=(rank, world), exchange=parallel.broadcast_object)
        t = p['text']
        q.put((rank, p['first'], p['n'], t.n, t.first, t.total, p['S'], p['R'], list(p['rg_to_int']),
               p['seq'].tobytes(), p['cseq'].tobytes(), p['qual'].tobytes(), p['meta'].tobytes()))
    finally:
        dist.destroy_process_group()


def test_ranks_index_only_their_shard_after_rank_0_scanned(tmp_path):
    """fastx.pack_pair with parallel.broadcast_object as the exchange, 3 gloo ranks: rank 0 scans the pair and hands
    out the plan, ranks 1 and 2 hold readers of their shard only; the shards concatenate to the single-process pack."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import oracle as O
    from kbbq import fastx
    n, world = 4000, 3
    seq, cseq, qual, meta = O.synth(0, n, n, 9, 50, 120, 4)
    names = O.synth_names(0, n, 4, with_rg=True)
    fa, fb = str(tmp_path / 'a.fq'), str(tmp_path / 'b.fq')
    O.write_fastq(fa, names, seq, qual, meta)
    O.write_fastq(fb, names, cseq, qual, meta)
    whole = fastx.pack_pair(fa, fb, True)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pack_worker, args=(r, world, port, fa, fb, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [g[3] for g in got] == [n] + [g[2] for g in got[1:]]          # rank 0 indexed everything, the others their shard
    assert all(g[4] == (0 if g[0] == 0 else g[1]) and g[5] == n for g in got)
    assert all((g[6], g[7], g[8]) == (whole['S'], whole['R'], list(whole['rg_to_int'])) for g in got)
    for k, name in ((9, 'seq'), (10, 'cseq'), (11, 'qual'), (12, 'meta')):
        assert b''.join(g[k] for g in got) == whole[name].tobytes(), name
