#!/usr/bin/env python3
"""
bench.py -- bases/s recalibrated (covariate accumulate + delta-Q solve + apply) on
synthetic 2x150 bp reads, device-resident, at N GPUs of one node.

    python bench.py --gpus N --steps K --warmup W

N = 1 runs in this process.  N > 1 without a launcher's environment: this process -- before it imports
anything that could touch a GPU -- starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
--master-addr 127.0.0.1 --master-port P bench.py ...` as a CHILD and relays its one JSON line; launched
by torch.distributed.run itself (RANK set) it is one rank of the job.  On a box with fewer GPUs than
ranks the ranks share the devices and talk over gloo: a functional rehearsal, flagged as such in `data`.

A "step" is one pass of the hot path over the rank's whole batch:
  K1 accumulate -> (N > 1: RCCL sum-allreduce of the count tables) -> delta-Q solve -> K2 apply.
Workload at N = 1: BASELINE.json configs[1] (50 M reads, 1 read group, Q0-41); weak scaling:
every rank holds --reads reads (configs[3] at N = 8).  Inputs are generated on the device and
are resident in HBM before the timed region.  Prints ONE JSON line on rank 0.

The line's `extra` object (N = 1 only, --no-extra skips it) carries the numbers of the other
configurations and of the rows next to the path, each measured in this same process after the timed
region: BASELINE config 3 (8 read groups) with its layout passes, the one-read-per-row layout, the
kernels of the truth-set / aligned-read rows (K4, K5, K6), the in-process FASTQ file path, and the CPU
figures (the reference's own measured rate from the committed golden, the oracle port on 1 and on all
host cores).
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, 'kbbq-py_amd'))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
READ_LEN = 150
KERNEL_SOURCES = ('kbbq_kernels.h', 'kbbq_kernels_v3.h', 'kbbq_k2_tile.h')      # where K1 / K2 live: what profiles/pmc_traffic.json is keyed to


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--reads', type=int, default=50_000_000, help='reads per GPU')
    ap.add_argument('--rgs', type=int, default=1)
    ap.add_argument('--cpu-sample', type=int, default=6_000_000, help='reads in the CPU baseline sample (0 = skip)')
    ap.add_argument('--layout', choices=('packed', 'pairs', 'reads'), default='packed',
                    help='device layout of the resident batch: mate-pair rows with 4-bit sequence planes, mate-pair rows '
                         '(304 B per 2 x 150 bp), or one read per row (2 x 160 B)')
    ap.add_argument('--no-extra', action='store_true', help='skip the `extra` object (other configurations, N = 1 only)')
    return ap.parse_args(argv)


# ---------------------------------------------------------------- launcher (N > 1 from a plain command line)
def free_port():
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def child_command(args, port):
    """The torch.distributed.run command line of the N-rank job (one rank per GPU)."""
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__),
           '--gpus', str(args.gpus), '--steps', str(args.steps), '--warmup', str(args.warmup),
           '--reads', str(args.reads), '--rgs', str(args.rgs), '--cpu-sample', str(args.cpu_sample),
           '--layout', args.layout]
    if args.no_extra:
        cmd.append('--no-extra')
    return cmd


def launch_children(args):
    """Parent of an N > 1 run started without a launcher: nothing GPU-side has been imported or initialised here;
    the job runs as a child process (never exec) and its single JSON line is relayed."""
    assert 'torch' not in sys.modules
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '4')
    proc = subprocess.run(child_command(args, free_port()), env=env, stdout=subprocess.PIPE)
    lines = [ln for ln in proc.stdout.decode(errors='replace').splitlines() if ln.startswith('{')]
    if lines:
        print(lines[-1], flush=True)
    return proc.returncode if proc.returncode else (0 if lines else 1)


# ---------------------------------------------------------------- CPU figures
def cpu_port_one_core(sample_reads):
    """The CPU oracle (oracle/, a scalar C port of the reference's algorithm) on a bounded
    sample of the same workload, 1 core.  Reported beside the GPU number; not the target."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import oracle as O
    seq, cseq, qual, meta = O.synth(0, sample_reads, sample_reads, 1)
    t0 = time.perf_counter()
    vectors = O.accumulate(seq, cseq, qual, meta, 1, READ_LEN)
    t1 = time.perf_counter()
    dqs = O.get_delta_qs(*vectors)
    t2 = time.perf_counter()
    O.apply(seq, qual, meta, vectors[0], *dqs)
    t3 = time.perf_counter()
    bases = sample_reads * READ_LEN
    return {'value': bases / (t3 - t0), 'unit': 'bases/s', 'cores': 1, 'kind': 'port',
            'sample': '%d synthetic 2x150 reads (%d bases), oracle C port: accumulate %.2fs + '
                      'solve %.2fs + apply %.2fs' % (sample_reads, bases, t1 - t0, t2 - t1, t3 - t2)}


def _port_worker(job):
    first, n, total = job
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import oracle as O
    seq, cseq, qual, meta = O.synth(first, n, total, 1)
    t0 = time.perf_counter()
    vectors = O.accumulate(seq, cseq, qual, meta, 1, READ_LEN)
    t1 = time.perf_counter()
    return [v.tolist() for v in vectors[5:9]], t1 - t0, first, n


def _apply_worker(job):
    first, n, total, meanq, dqs = job
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import numpy as np
    import oracle as O
    seq, cseq, qual, meta = O.synth(first, n, total, 1)
    t0 = time.perf_counter()
    O.apply(seq, qual, meta, np.asarray(meanq), *[np.asarray(d) for d in dqs])
    return time.perf_counter() - t0


def host_cores():
    """CPUs this process may use: the cgroup quota when there is one, else the affinity mask."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_port_all_cores(sample_reads):
    """The same port on every host core: reads are independent, the tables add (the reference itself is
    single-threaded; this is BASELINE.md's "restatement on all host cores" figure).  Shards are tallied in
    worker processes, the tables summed, solved once, and the shards applied in the same workers; the time
    is wall time of the two parallel phases plus the solve, generation excluded."""
    import multiprocessing as mp
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import oracle as O
    cores = host_cores()
    per = sample_reads // cores // 2 * 2
    jobs = [(k * per, per, cores * per) for k in range(cores)]
    with mp.get_context('spawn').Pool(cores) as pool:       # fresh children: nothing of this process (a profiler's preloads, a HIP context) is inherited
        t0 = time.perf_counter()
        parts = pool.map(_port_worker, jobs)
        t1 = time.perf_counter()
        tabs = [sum(np.asarray(p[0][i], dtype=np.int64) for p in parts) for i in range(4)]
        from kbbq import _solve                       # marginals + meanq of the summed tables (host NumPy)
        vectors = _solve.vectors_from_tables(*tabs)
        dqs = O.get_delta_qs(*vectors)
        t2 = time.perf_counter()
        ajobs = [(f, n, tot, vectors[0].tolist(), [d.tolist() for d in dqs]) for f, n, tot in jobs]
        t3 = time.perf_counter()
        ts = pool.map(_apply_worker, ajobs)
        t4 = time.perf_counter()
    gen = (t1 - t0) - max(p[1] for p in parts)          # workers generate their shard before the timed tally
    bases = cores * per * READ_LEN
    wall = max(p[1] for p in parts) + (t2 - t1) + max(ts)
    return {'value': bases / wall, 'unit': 'bases/s', 'cores': cores, 'kind': 'port',
            'sample': '%d synthetic 2x150 reads in %d worker processes: slowest tally %.2fs + solve %.2fs + slowest '
                      'apply %.2fs (generation %.1fs not counted)' % (cores * per, cores, max(p[1] for p in parts), t2 - t1, max(ts), gen)}


def reference_figure():
    """What the UNMODIFIED reference measured on BASELINE config 1 in the build container (it cannot travel to the GPU
    box): timing fields of tests/golden/c1_10k_1rg.json, written by oracle/gen_golden.py."""
    try:
        with open(os.path.join(ROOT, 'tests', 'golden', 'c1_10k_1rg.json')) as fh:
            t = json.load(fh)['reference_timing']
        return t
    except Exception:
        return None


# ---------------------------------------------------------------- helpers
def kernel_source_sha():
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        p = os.path.join(ROOT, 'kbbq-py_amd', 'csrc', name)
        if os.path.exists(p):
            with open(p, 'rb') as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_traffic(layout_key, kernel):
    """HBM bytes per base of `kernel` from the committed PMC passes (profiles/pmc_traffic.json: rocprofv3 --pmc
    FETCH_SIZE x 2 + WRITE_SIZE, separate passes) -- only when they were taken on THIS source of K1 / K2 (sha of
    csrc/kbbq_kernels.h + kbbq_kernels_v3.h + kbbq_k2_tile.h); otherwise None."""
    try:
        with open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')) as fh:
            pmc = json.load(fh)
        if pmc.get('kernel_source_sha') != kernel_source_sha():
            return None
        return pmc[layout_key][kernel]['hbm_bytes_per_base']
    except Exception:
        return None


def gbs(nbytes, ms):
    return nbytes / (ms * 1e-3) / 1e9


class Resident:
    """A device-resident synthetic batch in the layout the product would pick, with the cost of the layout passes."""

    def __init__(self, dev, torch, rank, n, world, R, layout, single_end=False):
        S = READ_LEN
        self.dev, self.torch, self.R, self.S, self.n = dev, torch, R, S, n
        batch = dev.ReadBatch.synthetic(rank * n, n, world * n, seed=1, nrg=R)
        if single_end:
            batch.meta.bitwise_and_(0x7FFFFFFF)      # no read is second in pair (compare_reads.py:304-306): single-end input
        self.rows = batch                    # one read per row, input order: what the packer hands over
        self.layout_ms = {}
        self.name = 'one read per row, pitch %d' % batch.pitch
        ev = lambda: torch.cuda.Event(enable_timing=True)
        if layout != 'reads':
            first = dev.lay_out(batch, R, S, packed=(layout == 'packed'))       # the process's first launches: code load, allocations
            del first
            a, b = ev(), ev()
            a.record()
            batch = dev.lay_out(batch, R, S, packed=(layout == 'packed'))
            b.record()
            torch.cuda.synchronize()
            self.layout_ms['lay_out'] = a.elapsed_time(b)
            self.name = batch.describe()
        elif R > 1:
            a, b = ev(), ev()
            a.record()
            batch = dev.group_by_rg(batch, R)
            b.record()
            torch.cuda.synchronize()
            self.layout_ms['lay_out'] = a.elapsed_time(b)
            self.name += ', rows grouped by read group'
        self.batch = batch
        self.out = torch.empty_like(batch.qual)
        self.tables = dev.Tables(R, 2 * S)

    def free_rows(self, check_reads=1_000_000):
        """Drop the input-order rows, keeping the character rows behind the first rows of the resident batch: after the
        timed region the output of those rows is checked against the persistent apply kernel on one read per row
        (`verify`) -- the layout the headline is measured on, against the kernel and layout it does not use."""
        torch, dev, b = self.torch, self.dev, self.batch
        self.check = None
        if b is not self.rows and check_reads:
            two = isinstance(b, dev.PairBatch)
            nrows = min(b.n, check_reads // 2 if two else check_reads)
            src = b.perm[:nrows] if b.perm is not None else torch.arange(nrows, device=b.seq.device)
            reads = torch.stack([2 * src, 2 * src + 1], 1).flatten() if two else src
            reads = reads[reads < self.rows.n]
            chk = dev.ReadBatch(int(reads.numel()), self.rows.pitch, with_corrected=False)
            chk.seq.copy_(self.rows.seq[reads]); chk.qual.copy_(self.rows.qual[reads]); chk.meta.copy_(self.rows.meta[reads])
            self.check = (chk, nrows, two)
        elif b is self.rows and check_reads and b.seg is None and not isinstance(b, dev.PairBatch):
            # character rows as they are (what a caller's rows get: since round 4 the short-lived K2): the first rows once more, for the
            # persistent kernel to look at
            nrows = min(b.n, check_reads)
            chk = dev.ReadBatch(nrows, b.pitch, with_corrected=False)
            chk.seq.copy_(b.seq[:nrows]); chk.qual.copy_(b.qual[:nrows]); chk.meta.copy_(b.meta[:nrows])
            self.check = (chk, nrows, False)
        self.rows = None
        torch.cuda.synchronize()
        torch.cuda.empty_cache()

    def verify(self):
        """True when the new qualities of the checked rows (self.out, as the last step left it) equal what the persistent
        kernel makes of the same reads on character rows with the LUT of the same tables; None when nothing was kept."""
        if self.check is None:
            return None
        torch, dev = self.torch, self.dev
        chk, nrows, two = self.check
        lut, shape = dev.solve_lut(self.tables)
        saved = os.environ.get('KBBQ_K2_TILE_CHARS')
        os.environ['KBBQ_K2_TILE_CHARS'] = '0'                  # the PERSISTENT kernel (k2v3_apply): another kernel and another layout than the timed one
        try:
            want = dev.apply(chk, lut, shape)
        finally:
            if saved is None:
                os.environ.pop('KBBQ_K2_TILE_CHARS', None)
            else:
                os.environ['KBBQ_K2_TILE_CHARS'] = saved
        got, S = self.out[:nrows], self.S
        if two:
            ok = torch.equal(got[:, :S], want[0::2, :S]) and torch.equal(got[:want.shape[0] // 2, S + 1:2 * S + 1], want[1::2, :S])
        else:
            ok = torch.equal(got[:, :S], want[:, :S])
        return bool(ok) and bool((want[:, :S] != chk.qual[:, :S]).any().item())      # and the kernel did change qualities


def timed_steps(torch, dist, use_dist, dev, parallel, res, steps, warmup, rehearse, restore_order=False):
    ctx = dev.context()
    ar_events = []

    def step(timed):
        res.tables.buf.zero_()
        dev.accumulate(res.batch, res.tables, check=False)
        if use_dist:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            parallel.allreduce_tables(res.tables.buf)
            b.record()
            if timed:
                ar_events.append((a, b))
        lut, shape = dev.solve_lut(res.tables, check=False, reuse=True)
        dev.apply(res.batch, lut, shape, out=res.out, check=False, restore_order=restore_order)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step(False)
    ctx.status()                       # raises if a kernel flagged bad input
    ctx.kernel_ms(0, reset=True); ctx.kernel_ms(1, reset=True)
    ctx.timing(True)
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(True)
    fence()
    elapsed = time.perf_counter() - t0
    ctx.timing(False)
    ctx.status()
    k1_ms, k1_n = ctx.kernel_ms(0)
    k2_ms, k2_n = ctx.kernel_ms(1)
    ar_ms = sum(a.elapsed_time(b) for a, b in ar_events) / max(len(ar_events), 1) if ar_events and not rehearse else None
    return elapsed, k1_ms / max(k1_n, 1), k2_ms / max(k2_n, 1), k1_n, k2_n, ar_ms


def kernel_entry(avg_ms, launches, bases, bytes_per_base=3):
    rate = gbs(bytes_per_base * bases, avg_ms)
    return {'avg_ms': avg_ms, 'launches': launches, 'GB/s': rate, 'frac': rate / HBM_PEAK_GBS, 'bytes_per_base': bytes_per_base}


# ---------------------------------------------------------------- the `extra` object (N = 1)
def extra_config3(torch, dev, parallel, n, steps, warmup):
    """BASELINE config 3: n reads, 8 read groups.  `value` as the headline (layout pass outside the step, like the
    generation: in the file path the packer writes this layout itself); `layout_inclusive` is the figure for a caller
    who hands over input-order rows on the device: the one native pass into the layout (read-group counting sort +
    k7_lay_out: pairs packed, rows gathered by read-group segment, sequences as nibbles) plus a step whose K2 stores
    through the permutation, straight back into input order (no pass afterwards)."""
    res = Resident(dev, torch, 0, n, 1, 8, 'packed')
    res.free_rows()
    elapsed, k1, k2, n1, n2, _ = timed_steps(torch, None, False, dev, parallel, res, steps, warmup, False)
    verified = res.verify()
    elapsed_r, k1r, k2r, _, n2r, _ = timed_steps(torch, None, False, dev, parallel, res, steps, 1, False, restore_order=True)
    bases = n * READ_LEN
    step_ms, step_r_ms = elapsed / steps * 1e3, elapsed_r / steps * 1e3
    lay = res.layout_ms.get('lay_out', 0.0)
    unpack_ms = None
    if isinstance(res.batch, dev.PairBatch):              # pair rows of new qualities -> one read per row (callers that want rows)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        rows = res.batch.unpack(res.out); del rows
        a.record(); rows = res.batch.unpack(res.out); b.record()
        torch.cuda.synchronize()
        unpack_ms = a.elapsed_time(b)
        del rows
    return {'workload': '%d synthetic 2x150 bp reads, 8 read groups (BASELINE config 3)' % n, 'layout': res.name, 'verified': verified,
            'value': bases * steps / elapsed, 'unit': 'bases/s', 'ms_per_step': step_ms,
            'k1_accumulate': kernel_entry(k1, n1, bases), 'k2_apply': kernel_entry(k2, n2, bases),
            'host_solve_and_sync_ms': step_ms - k1 - k2,
            'lay_out_ms': lay, 'unpack_pairs_ms': unpack_ms, 'k2_apply_storing_through_perm': kernel_entry(k2r, n2r, bases),
            'layout_inclusive': {'value': bases / ((step_r_ms + lay) * 1e-3), 'unit': 'bases/s', 'ms': step_r_ms + lay,
                                 'note': 'input-order rows on the device -> layout pass (%.2f ms) + one step whose K2 stores '
                                         'back in input order (%.2f ms); the output is mate-pair rows in input order (the writer reads '
                                         'them as they are; unpack_pairs_ms is the pass to one read per row for callers that want it)' % (lay, step_r_ms)}}


def extra_layout(torch, dev, parallel, n, steps, warmup, layout, single_end=False):
    res = Resident(dev, torch, 0, n, 1, 1, layout, single_end=single_end)
    res.free_rows()
    elapsed, k1, k2, n1, n2, _ = timed_steps(torch, None, False, dev, parallel, res, steps, warmup, False)
    bases = n * READ_LEN
    return {'workload': ('%d synthetic single-end 150 bp reads, 1 read group' if single_end else '%d synthetic 2x150 bp reads, 1 read group') % n,
            'layout': res.name, 'verified': res.verify(),
            'value': bases * steps / elapsed, 'unit': 'bases/s', 'ms_per_step': elapsed / steps * 1e3,
            'k1_accumulate': kernel_entry(k1, n1, bases), 'k2_apply': kernel_entry(k2, n2, bases)}


def extra_mixed_lengths(torch, dev, n, steps, warmup, lo=36, hi=300):
    """BASELINE config 5, the recalibration half, device-resident: n reads of lo..hi bases in the length bands the file
    path cuts (kbbq/fastx.py BAND_CLASSES: every band at its own pitch), each band in the layout the product picks for
    it (kbbq.fastx._fill_bands: 4-bit planes, one read per row), count tables of 2 x hi columns.
    A step = K1 over every band (into a band's own tables, added to the file's: recalibrate._tally_local) -> solve ->
    K2 over every band."""
    from kbbq import fastx, recalibrate
    bands, at = [], lo
    for c in fastx.BAND_CLASSES:
        if c >= at:
            bands.append((at, min(c, hi)))
            at = min(c, hi) + 1
        if at > hi:
            break
    per = max(2, (n // len(bands)) & ~1)
    tables, part = dev.Tables(1, 2 * hi), dev.Tables(1, 2 * hi)
    items, bases, padded = [], 0, 0
    for k, (blo, bhi) in enumerate(bands):
        batch = dev.ReadBatch.synthetic(k * per, per, per * len(bands), seed=1, len_lo=blo, len_hi=bhi)
        st = dev.meta_stats(batch)
        rows = dev.lay_out(batch, 1, st['longest'], packed=st['longest'] <= dev.PACKED_READS, pairs=None if st['longest'] == hi else False, stats=st)
        bases += int(batch.lengths_host().sum())
        padded += rows.n * rows.pitch
        items.append({'rows': rows, 'S': st['longest'], 'Smin': st['shortest'], 'out': torch.empty_like(rows.qual)})
        del batch
    torch.cuda.empty_cache()
    ctx = dev.context()

    triples = [(it['rows'], it['S'], it['Smin']) for it in items]
    outs = [it['out'] for it in items]

    def step_per_band():
        tables.buf.zero_()
        for it in items:
            part.buf.zero_()
            dev.accumulate(it['rows'], part, check=False, s_band=it['S'], s_min=it['Smin'])
            tables.add(part)
        lut, shape = dev.solve_lut(tables, check=False, reuse=True)
        for it in items:
            dev.apply(it['rows'], lut, shape, out=it['out'], check=False)

    def step():                             # what the file path does: one K1 launch and one K2 launch over all bands
        tables.buf.zero_()
        part.buf.zero_()
        dev.accumulate_bands(triples, part, check=False)
        tables.add(part)
        lut, shape = dev.solve_lut(tables, check=False, reuse=True)
        dev.apply_bands(triples, lut, shape, outs=outs, check=False)
    # the launch-per-band form first: its tables and new qualities are what the merged launches must reproduce
    step_per_band()
    ctx.status()
    want_tables = tables.buf.clone()
    want_outs = [o.clone() for o in outs]
    ctx.kernel_ms(0, reset=True); ctx.kernel_ms(1, reset=True)
    ctx.timing(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_per_band()
    torch.cuda.synchronize()
    per_band_s = time.perf_counter() - t0
    ctx.timing(False)
    pb1, pb1n = ctx.kernel_ms(0, reset=True)
    pb2, pb2n = ctx.kernel_ms(1, reset=True)
    for o in outs:
        o.zero_()
    for _ in range(warmup):
        step()
    ctx.status()
    ctx.kernel_ms(0, reset=True); ctx.kernel_ms(1, reset=True)
    ctx.timing(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ctx.timing(False)
    ctx.status()
    k1_ms, k1_n = ctx.kernel_ms(0)
    k2_ms, k2_n = ctx.kernel_ms(1)
    same = bool(torch.equal(tables.buf, want_tables)) and all(bool(torch.equal(a, b)) for a, b in zip(outs, want_outs))
    return {'workload': '%d synthetic reads of %d-%d bases in %d length bands of %d reads (BASELINE config 5, recalibration half), '
                        '1 read group, tables of %d cycle columns' % (per * len(bands), lo, hi, len(bands), per, 2 * hi),
            'launches': 'ONE K1 launch and ONE K2 launch over all bands (kbbq_accumulate_bands_dev / kbbq_apply_bands_dev: every band on '
                        'its share of the workgroups, its own pitch, LDS geometry and pitch-narrowed LUT)',
            'verified': same, 'verified_how': 'count tables and every new quality byte == the launch-per-band form on the same bands',
            'launch_per_band': {'value': bases * steps / per_band_s, 'ms_per_step': per_band_s / steps * 1e3,
                                'k1_accumulate_all_bands': kernel_entry(pb1 / steps, pb1n, bases),
                                'k2_apply_all_bands': kernel_entry(pb2 / steps, pb2n, bases)},
            'layout': '; '.join('%d-%d: %s' % (b[0], b[1], it['rows'].describe()) for b, it in zip(bands, items)),
            'bases_per_step': bases, 'padded_row_bytes_per_plane': padded,
            'value': bases * steps / elapsed, 'unit': 'bases/s', 'ms_per_step': elapsed / steps * 1e3,
            'k1_accumulate_all_bands': kernel_entry(k1_ms / steps, k1_n, bases),
            'k2_apply_all_bands': kernel_entry(k2_ms / steps, k2_n, bases),
            'outside_the_kernels_ms': (elapsed * 1e3 - k1_ms - k2_ms) / steps}


def extra_aligned(torch, dev, n=4_000_000, L=150, G=200_000_000, ins=0.05, reps=5):
    """K4 find_errors / K5 count_q / K6 canonical_reads on synthetic aligned reads built as arrays: n reads x L bases
    against a random genome, `ins` of the reads with a 2-base insertion (3 CIGAR operations), half of them on the
    reverse strand.  Algorithmic bytes per base: K4 3 read + 2 written, K5 3 read, K6 4 + 3."""
    from kbbq import _native as N
    pitch = (L + 15) // 16 * 16
    g = torch.randint(0, 4, (G,), dtype=torch.uint8, device='cuda')
    genome = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device='cuda')[g.long()]
    del g
    mask = (torch.rand(G, device='cuda') < 0.01).to(torch.uint8)
    fused = genome | (mask << 7)
    del mask
    start = torch.randint(0, G - L, (n,), dtype=torch.int64, device='cuda')
    start[:64] = G - L
    idx = (start[:, None] + torch.arange(pitch, device='cuda')[None, :]).clamp_(max=G - 1)
    seq = genome[idx]
    del idx, genome
    seq = torch.where(torch.rand((n, pitch), device='cuda') < 0.01, torch.tensor(65, dtype=torch.uint8, device='cuda'), seq)
    lens = torch.full((n,), L, dtype=torch.int32, device='cuda')
    has_ins = torch.rand(n, device='cuda') < ins
    ref_len = torch.where(has_ins, L - 2, L).to(torch.int32)
    cig_n = torch.where(has_ins, 3, 1).to(torch.int32)
    cig_off = torch.cumsum(cig_n, 0).to(torch.int32) - cig_n
    cigar = torch.zeros(int(cig_n.sum()), dtype=torch.int32, device='cuda')
    o = cig_off.long()
    cigar[o[~has_ins]] = (L << 4) | 0
    cigar[o[has_ins]] = (50 << 4) | 0
    cigar[o[has_ins] + 1] = (2 << 4) | 1
    cigar[o[has_ins] + 2] = ((L - 52) << 4) | 0
    flip = (torch.rand(n, device='cuda') < 0.5).to(torch.uint8)
    err = torch.zeros((n, pitch), dtype=torch.uint8, device='cuda')
    skip = torch.zeros_like(err)
    qual = torch.randint(2, 42, (n, pitch), dtype=torch.uint8, device='cuda')
    counts = torch.zeros(512, dtype=torch.int64, device='cuda')
    oq = qual + 33
    noflip = torch.zeros_like(flip)
    clip = torch.full((n,), L << 16, dtype=torch.int32, device='cuda')
    clip[torch.rand(n, device='cuda') < 0.2] = 5 | ((L - 5) << 16)
    trim = torch.zeros(n, dtype=torch.int32, device='cuda')
    trim[torch.rand(n, device='cuda') < 0.05] = (L - 20) | (L << 16)
    flags = torch.randint(0, 4, (n,), device='cuda', dtype=torch.int32)
    batch = dev.ReadBatch(n, pitch, with_corrected=True)
    packed = dev.ReadBatch(n, pitch, with_corrected=True, nib=True)      # what gatk/bqsr.py puts between K6 and K1
    tables = dev.Tables(1, 2 * L)
    ctx, lib = dev.context(), N.load()

    def k4(fl, two_planes=False):
        N.check(lib.kbbq_find_errors_dev(ctx.handle, N.ptr(seq), N.ptr(lens), n, pitch, N.ptr(start), N.ptr(ref_len),
                                         N.ptr(cig_off), N.ptr(cig_n), N.ptr(cigar), N.ptr(fused), None, G, N.ptr(fl),
                                         N.ptr(err), N.ptr(skip) if two_planes else None))

    def k5(two_planes=False):
        N.check(lib.kbbq_count_q_dev(ctx.handle, N.ptr(qual), N.ptr(err), N.ptr(skip) if two_planes else None, N.ptr(lens), n, pitch, 0, N.ptr(counts)))

    def k6(two_planes=False, to=packed):
        N.check(lib.kbbq_canonical_reads_rows_dev(ctx.handle, N.ptr(seq), N.ptr(oq), N.ptr(err), N.ptr(skip) if two_planes else None,
                                                  N.ptr(lens), N.ptr(clip), N.ptr(trim), N.ptr(flags), n, pitch, L, 6, 6,
                                                  N.ROWS_NIBBLES if to.nib else 0,
                                                  N.ptr(to.seq), N.ptr(to.cseq), N.ptr(to.qual), N.ptr(to.meta)))

    def k1(of=packed):
        dev.accumulate(of, tables, 6, check=False, dinuc_minscore=6)

    def k61():                              # K6 fused into K1: the tally straight from the reads as aligned
        N.check(lib.kbbq_accumulate_aligned_dev(ctx.handle, N.ptr(seq), N.ptr(oq), N.ptr(err), N.ptr(clip), N.ptr(trim), N.ptr(flags), n, pitch, L,
                                                1, 6, 6, N.ptr(tables.buf)))

    plane = torch.empty((n, pitch), dtype=torch.uint8, device='cuda')

    def k461():                             # K4 folded in as well: reads of one M operation never meet a plane of flags
        N.check(lib.kbbq_tally_aligned_dev(ctx.handle, N.ptr(seq), N.ptr(oq), N.ptr(lens), n, pitch, L, N.ptr(start), N.ptr(ref_len),
                                           N.ptr(cig_off), N.ptr(cig_n), N.ptr(cigar), N.ptr(fused), G, N.ptr(clip), N.ptr(trim), N.ptr(flags),
                                           N.ptr(plane), 1, 6, 6, N.ptr(tables.buf)))

    out = {'workload': '%d aligned reads x %d bp, %d Mb random genome, %.0f %% of the reads with a 2-base insertion, half '
                       'reverse-strand; HIP events on the launch stream, %d launches each' % (n, L, G // 1000000, ins * 100, reps),
           'bytes_per_base': 'algorithmic, by the arrays of the reference: K4 3 read (read, reference, site mask) + 2 written (errors, '
                             'skips); K5 3 read; K6 4 read + 3 written.  The product keeps errors and skips in ONE plane of flags between '
                             'these kernels (and the site mask in bit 7 of the reference bytes): each kernel moves 1 B/base less than its '
                             'algorithmic count; the *_two_planes entries are the form with separate error / skip planes.  K6 hands K1 '
                             '4-bit sequence planes (2 B/base written / read instead of 3); the *_character_planes entries are the form '
                             'with one byte per base (taken when a read holds a letter outside ACGTN)'}
    bases = n * L
    for name, fn, bpb in (('k4_find_errors', lambda: k4(flip), 5), ('k5_count_q', k5, 3),
                          ('k4_find_errors_tally', lambda: k4(noflip), 5), ('k6_canonical_reads', k6, 7),
                          ('k1_on_canonical_reads', k1, 3), ('k61_fused_tally', k61, 3), ('k461_whole_tally_one_pass', k461, 3),
                          ('k6_canonical_reads_character_planes', lambda: k6(False, batch), 7),
                          ('k1_on_canonical_reads_character_planes', lambda: k1(batch), 3),
                          ('k4_find_errors_two_planes', lambda: k4(flip, True), 5), ('k5_count_q_two_planes', lambda: k5(True), 3),
                          ('k6_canonical_reads_two_planes', lambda: k6(True, batch), 7)):
        fn()
        torch.cuda.synchronize()
        evs = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record()
            evs.append((a, b))
        torch.cuda.synchronize()
        ms = sum(a.elapsed_time(b) for a, b in evs) / reps
        out[name] = kernel_entry(ms, reps, bases, bpb)
    ctx.status()
    # the fused kernel counts what K6 -> K1 counts (same K4 flags, same reads): checked here on the bench's own data
    k4(noflip); tables.buf.zero_(); k6(); k1(); want = tables.buf.clone()
    tables.buf.zero_(); k61()
    ctx.status()
    out['k61_fused_tally']['verified'] = bool(torch.equal(tables.buf, want)) and int(want.sum()) > 0
    tables.buf.zero_(); plane.fill_(0xA5); k461()
    ctx.status()
    out['k461_whole_tally_one_pass']['verified'] = bool(torch.equal(tables.buf, want))
    out['k461_whole_tally_one_pass']['note'] = ('wall time of kbbq_tally_aligned_dev (HIP events around the call): the records / classification pass, '
                                                'K4 over the reads that are not one M operation (here %.0f %% insertion reads and the reads at the genome\'s end) '
                                                'and the tally kernel that compares the other reads with the reference itself' % (ins * 100))
    out['whole_tally_ms'] = {'k4_k6_k1': out['k4_find_errors_tally']['avg_ms'] + out['k6_canonical_reads']['avg_ms'] + out['k1_on_canonical_reads']['avg_ms'],
                             'k4_fused': out['k4_find_errors_tally']['avg_ms'] + out['k61_fused_tally']['avg_ms'],
                             'one_pass': out['k461_whole_tally_one_pass']['avg_ms'],
                             'note': 'gatk.bqsr.bam_to_bqsr_covariates on the device: K4 (flags) then K6 -> K1 (canonical reads written and read back: 3 + 2 + 2 '
                                     'B/base), K4 then the fused kernel (kbbq_accumulate_aligned_dev: 3 + 3 B/base), or everything in one pass '
                                     '(kbbq_tally_aligned_dev: 3 B/base for reads of one M operation; what gatk.bqsr runs)'}
    return out


def extra_file_path(torch, dev, n=8_000_000):
    """FASTQ text in -> FASTQ text out, in this (warm) process: kbbq.recalibrate.recalibrate_fastq on a synthetic
    2x150 bp pair written to the box's /tmp, output into a file; stage times from the path's own KBBQ_TIMING trace."""
    import numpy as np
    from kbbq import recalibrate, _trace
    tmp = os.environ.get('TMPDIR', '/tmp')
    fa, fb, fo = (os.path.join(tmp, 'kbbq_bench_%d_%s.fq' % (os.getpid(), x)) for x in 'abo')
    batch = dev.ReadBatch.synthetic(0, n, n, seed=1)
    seq, cseq, qual = (getattr(batch, p)[:n, :READ_LEN].cpu().numpy() for p in ('seq', 'cseq', 'qual'))
    del batch
    t0 = time.perf_counter()
    try:
        for path, plane in ((fa, seq), (fb, cseq)):
            rec = np.empty((n, 1 + 12 + 1 + 150 + 3 + 150 + 1), dtype=np.uint8)     # fixed-width names r%09d/1
            ids = np.arange(n)
            digits = ((ids >> 1)[:, None] // 10 ** np.arange(8, -1, -1)[None, :] % 10 + 48).astype(np.uint8)
            rec[:, 0] = ord('@'); rec[:, 1] = ord('r'); rec[:, 2:11] = digits; rec[:, 11] = ord('/')
            rec[:, 12] = 49 + (ids & 1); rec[:, 13] = 10
            rec[:, 14:164] = plane; rec[:, 164] = 10; rec[:, 165] = ord('+'); rec[:, 166] = 10
            rec[:, 167:317] = qual; rec[:, 317] = 10
            rec.tofile(path)
            del rec
        write_s = time.perf_counter() - t0
        sys.stdout.flush()
        saved = os.dup(1)
        fd = os.open(fo, os.O_WRONLY | os.O_CREAT | os.O_TRUNC)
        os.dup2(fd, 1)
        runs = []
        try:
            for rep in range(2):                              # the better of two: the first call of a process also page-locks its staging buffers
                os.lseek(1, 0, os.SEEK_SET)
                os.ftruncate(1, 0)
                _trace.collect(True)
                t0 = time.perf_counter()
                recalibrate.recalibrate_fastq([fa, fb])
                sys.stdout.flush()
                runs.append((time.perf_counter() - t0, _trace.collect(False)))
        finally:
            _trace.collect(False)
            os.dup2(saved, 1)
            os.close(fd)
            os.close(saved)
        wall, stages = min(runs, key=lambda r: r[0])
        size = os.path.getsize(fo)
        bands = list(recalibrate.LAST_RUN.get('bands', []))
        # the same files in constant device memory (kbbq/_stream.py: 256 MB of slabs instead of the whole shard) and through two
        # pipes (`-f <(cat A) <(cat B)`: sequential segments, file A spooled for pass 2) -- round 4; the bytes are compared
        import hashlib

        def digest(path):
            h = hashlib.sha256()
            with open(path, 'rb') as fh:
                for blk in iter(lambda: fh.read(1 << 24), b''):
                    h.update(blk)
            return h.hexdigest()
        want = digest(fo)
        other = {}
        for key, env, pipes in (('streamed_256M', {'KBBQ_DEVICE_BUDGET': '256M'}, False), ('pipes', {}, True)):
            os.remove(fo)
            saved_env = {k: os.environ.get(k) for k in env}
            os.environ.update(env)
            feeders, inputs = [], [fa, fb]
            try:
                if pipes:
                    inputs = [os.path.join(tmp, 'kbbq_bench_%d_fifo_%s' % (os.getpid(), x)) for x in 'ab']
                    for src, fifo in zip((fa, fb), inputs):
                        os.mkfifo(fifo)
                        feeders.append(subprocess.Popen('cat %s > %s' % (src, fifo), shell=True))
                _trace.collect(True)
                t0 = time.perf_counter()
                recalibrate.recalibrate_fastq(inputs, output=fo)
                w = time.perf_counter() - t0
                other[key] = {'wall_s': w, 'value': n * READ_LEN / w, 'unit': 'bases/s', 'stages_s': _trace.collect(False),
                              'same_bytes_as_resident': digest(fo) == want, 'run': dict(recalibrate.LAST_RUN.get('streamed') or {})}
            except Exception as e:                # noqa: BLE001
                _trace.collect(False)
                other[key] = {'error': '%s: %s' % (type(e).__name__, e)}
            finally:
                for f in feeders:
                    f.wait()
                for k, v in saved_env.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v
                if pipes:
                    for fifo in inputs:
                        if os.path.exists(fifo):
                            os.remove(fifo)
    finally:
        for p in (fa, fb, fo):
            if os.path.exists(p):
                os.remove(p)
    bases = n * READ_LEN
    return {'workload': '%d synthetic 2x150 bp reads as two FASTQ files (%.1f GB each) -> recalibrated FASTQ file (%.1f GB), '
                        'in-process, device warm' % (n, n * 318 / 1e9, size / 1e9),
            'value': bases / wall, 'unit': 'bases/s', 'wall_s': wall, 'walls_of_both_runs_s': [r[0] for r in runs], 'stages_s': stages, 'bands': bands,
            'h2d_bytes_per_base': sum(b['h2d_bytes'] for b in bands) / bases if bands else None,
            'input_written_in_s': write_s,
            'streamed_within_256MB_of_device_memory': other.get('streamed_256M'), 'through_two_pipes': other.get('pipes')}


def extra_compressed_inputs(torch, dev, n=1_000_000):
    """The same text as a `.fq.gz` pair (one gzip member per file, level 1 for the bench's own time): recalibrate_fastq in this
    process on the plain pair and on the compressed pair -- inflated on all host threads (csrc/parallel_gunzip.cpp) and, for
    comparison, by zlib on one thread per file (KBBQ_PGZ_MIN_BYTES / KBBQ_LIBDEFLATE switch the faster paths off)."""
    import hashlib
    import zlib
    import numpy as np
    from kbbq import recalibrate
    tmp = os.environ.get('TMPDIR', '/tmp')
    fa, fb, fo = (os.path.join(tmp, 'kbbq_benchz_%d_%s.fq' % (os.getpid(), x)) for x in 'abo')
    batch = dev.ReadBatch.synthetic(0, n, n, seed=1)
    seq, cseq, qual = (getattr(batch, p)[:n, :READ_LEN].cpu().numpy() for p in ('seq', 'cseq', 'qual'))
    del batch
    made = []
    try:
        sizes = {}
        for path, plane in ((fa, seq), (fb, cseq)):
            rec = np.empty((n, 318), dtype=np.uint8)
            ids = np.arange(n)
            digits = ((ids >> 1)[:, None] // 10 ** np.arange(8, -1, -1)[None, :] % 10 + 48).astype(np.uint8)
            rec[:, 0] = ord('@'); rec[:, 1] = ord('r'); rec[:, 2:11] = digits; rec[:, 11] = ord('/')
            rec[:, 12] = 49 + (ids & 1); rec[:, 13] = 10
            rec[:, 14:164] = plane; rec[:, 164] = 10; rec[:, 165] = ord('+'); rec[:, 166] = 10
            rec[:, 167:317] = qual; rec[:, 317] = 10
            raw = rec.tobytes()
            del rec
            with open(path, 'wb') as fh:
                fh.write(raw)
            z = zlib.compressobj(1, zlib.DEFLATED, 31)
            with open(path + '.gz', 'wb') as fh:
                fh.write(z.compress(raw)); fh.write(z.flush())
            made += [path, path + '.gz']
            sizes[os.path.basename(path)] = {'text_bytes': len(raw), 'gzip_bytes': os.path.getsize(path + '.gz')}
            del raw

        def run(a, b, env):
            saved = {k: os.environ.get(k) for k in env}
            os.environ.update(env)
            try:
                best = None
                for _ in range(2):
                    if os.path.exists(fo):
                        os.remove(fo)
                    t0 = time.perf_counter()
                    recalibrate.recalibrate_fastq([a, b], output=fo)
                    wall = time.perf_counter() - t0
                    best = wall if best is None else min(best, wall)
                h = hashlib.sha256()
                with open(fo, 'rb') as fh:
                    for blk in iter(lambda: fh.read(1 << 24), b''):
                        h.update(blk)
                return best, h.hexdigest()[:16]
            finally:
                for k, v in saved.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v
        plain, sha0 = run(fa, fb, {})
        wide, sha1 = run(fa + '.gz', fb + '.gz', {})
        one, sha2 = run(fa + '.gz', fb + '.gz', {'KBBQ_PGZ_MIN_BYTES': str(1 << 60)})
        bases = n * READ_LEN
        return {'workload': '%d synthetic 2x150 bp reads as plain text and as one gzip member per file (level 1)' % n, 'files': sizes,
                'plain_text': {'wall_s': plain, 'value': bases / plain, 'unit': 'bases/s'},
                'gzip_on_all_host_threads': {'wall_s': wide, 'value': bases / wide, 'unit': 'bases/s',
                                             'how': 'csrc/parallel_gunzip.cpp: chunks of the DEFLATE stream decoded side by side against unknown windows, resolved in order'},
                'gzip_on_one_thread_per_file': {'wall_s': one, 'value': bases / one, 'unit': 'bases/s', 'how': 'libdeflate (zlib without it), KBBQ_PGZ_MIN_BYTES=2^60'},
                'same_output_sha256': sha0 == sha1 == sha2}
    finally:
        for p in made + [fo]:
            if os.path.exists(p):
                os.remove(p)


def build_extra(torch, dev, parallel, args, headline_layout):
    extra, n = {}, args.reads
    small = max(args.steps // 2, 3)
    for key, fn in (('config3_8rg', lambda: extra_config3(torch, dev, parallel, n, small, 1)),
                    ('layout_pairs', lambda: extra_layout(torch, dev, parallel, n, small, 1, 'pairs')),
                    ('layout_reads', lambda: extra_layout(torch, dev, parallel, n, small, 1, 'reads')),
                    ('single_end_150', lambda: extra_layout(torch, dev, parallel, n, small, 1, 'packed', single_end=True)),
                    ('config5_mixed_lengths', lambda: extra_mixed_lengths(torch, dev, min(20_000_000, n), small, 1)),
                    ('aligned_read_kernels', lambda: extra_aligned(torch, dev, n=min(16_000_000, n), G=min(200_000_000, 50 * n))),
                    ('file_path', lambda: extra_file_path(torch, dev, n=min(8_000_000, n))),
                    ('compressed_inputs', lambda: extra_compressed_inputs(torch, dev, n=min(1_000_000, n)))):
        if key == 'layout_' + headline_layout:
            continue
        t0 = time.perf_counter()
        try:
            extra[key] = fn()
        except Exception as e:                # noqa: BLE001 -- an extra never takes the headline down
            extra[key] = {'error': '%s: %s' % (type(e).__name__, e)}
        extra[key]['took_s'] = round(time.perf_counter() - t0, 2)
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    src = extra.get('layout_reads')
    if src and 'error' not in src:
        extra['from_input_order_rows'] = dict(src, note='what kbbq.recalibrate does for rows as a caller holds them (one character row per read, '
                                                        'input order, on the device): K1 and K2 run on those rows as they are, no layout pass, no '
                                                        'unpack -- a device pass into the mate-pair / 4-bit layout costs more than it saves for ONE '
                                                        'accumulate + apply; the file path never has such rows: its packer writes the layout of the headline')
    return extra


# ---------------------------------------------------------------- one rank
def run_rank(args):
    if int(os.environ.get('WORLD_SIZE', '1')) != args.gpus:
        raise SystemExit('bench.py: --gpus %d but the launcher started %s ranks' % (args.gpus, os.environ.get('WORLD_SIZE', '1')))
    # stdout carries exactly ONE JSON line: library banners (RCCL prints its version to stdout under
    # NCCL_DEBUG=VERSION) are sent to stderr until the result is printed
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    # the CPU figures come FIRST: the all-core leg forks worker processes, which must not inherit a HIP context or the
    # runtime's threads (nothing GPU-side has been imported or initialised yet)
    cpu_base = None
    if world == 1 and rank == 0 and args.cpu_sample > 0:
        assert 'torch' not in sys.modules
        cpu_base = cpu_port_one_core(args.cpu_sample)
        if not args.no_extra:
            try:
                cpu_base['all_cores'] = cpu_port_all_cores(args.cpu_sample * 2)
            except Exception as e:        # noqa: BLE001
                cpu_base['all_cores'] = {'error': '%s: %s' % (type(e).__name__, e)}

    import torch
    import torch.distributed as dist
    local = int(os.environ.get('LOCAL_RANK', '0'))
    use_dist = 'RANK' in os.environ                 # launched by torch.distributed.run (any N, also 1)
    # fewer devices than ranks (a one-GPU box): every rank on the devices that exist, gloo instead of RCCL --
    # a functional rehearsal of the N > 1 path; its numbers mean nothing (KBBQ_BENCH_REHEARSE=gloo forces it)
    ndev = max(torch.cuda.device_count(), 1)
    rehearse = use_dist and (os.environ.get('KBBQ_BENCH_REHEARSE', '') == 'gloo' or ndev < world)
    if rehearse:
        local = local % ndev
    torch.cuda.set_device(local)
    if use_dist:
        if rehearse:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))

    from kbbq import _device as dev
    from kbbq import parallel
    host_binding = parallel.bind_host_threads(local) if use_dist else None     # threads = CPUs / LOCAL_WORLD_SIZE, on the GPU's NUMA node

    n, R, S = args.reads, args.rgs, READ_LEN
    res = Resident(dev, torch, rank, n, world, R, args.layout)
    res.free_rows()
    elapsed, k1_avg, k2_avg, k1_n, k2_n, ar_ms = timed_steps(torch, dist, use_dist, dev, parallel, res, args.steps,
                                                              args.warmup, rehearse)
    t = torch.tensor([elapsed], dtype=torch.float64, device='cpu' if rehearse or not use_dist else 'cuda')
    per_rank_ms = [elapsed / args.steps * 1e3]
    if use_dist:
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        per_rank_ms = [float(x.item()) / args.steps * 1e3 for x in every]      # stragglers show here
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    ranks_seen = dist.get_world_size() if use_dist else 1
    backend = dist.get_backend() if use_dist else None
    verified = res.verify()
    verified_ranks = [verified]
    if use_dist:
        # EVERY rank checks the output of its own batch; the line carries the AND over the ranks (None: nothing was kept to check)
        every = [None] * world
        dist.all_gather_object(every, verified)
        verified_ranks = every
        verified = None if any(v is None for v in every) else all(every)
    layout_name = res.name
    layout_pass_ms = res.layout_ms.get('lay_out')
    layout_key = res.batch.layout_key() if hasattr(res.batch, 'layout_key') else ('reads' if args.layout == 'reads' else 'pairs')
    del res
    torch.cuda.synchronize()
    torch.cuda.empty_cache()

    if rank == 0:
        bases_per_rank = n * S
        total_bases = bases_per_rank * world * args.steps
        # algorithmic bytes per launch: K1 reads seq+cseq+qual = 3 B/base;
        # K2 reads seq+qual and writes qual = 3 B/base
        k1 = kernel_entry(k1_avg, k1_n, bases_per_rank)
        k2 = kernel_entry(k2_avg, k2_n, bases_per_rank)
        dom, domk = ('k1_accumulate', k1) if k1_avg >= k2_avg else ('k2_apply', k2)
        # bytes the resident layout itself holds per base (4-bit planes: half a byte per sequence base; padding not counted)
        nibs = 'nib' in layout_key
        layout_bytes = {'k1_accumulate': 2.0 if nibs else 3.0, 'k2_apply': 2.5 if nibs else 3.0}
        per_base = pmc_traffic(layout_key, dom)
        for name, k in (('k1_accumulate', k1), ('k2_apply', k2)):
            pb = pmc_traffic(layout_key, name)
            k['hbm_bytes_per_base_pmc'] = pb
            k['hbm_GB/s_pmc'] = None if pb is None else gbs(pb * bases_per_rank, k['avg_ms'])
        step_ms = elapsed / args.steps * 1e3
        res_line = {
            'metric': 'bases/sec recalibrated (2x150 bp)', 'value': total_bases / elapsed,
            'unit': 'bases/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': step_ms, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'u8', 'data': 'synthetic' + (' (gloo rehearsal on shared devices, not a measurement)' if rehearse else ''),
            'config': {'workload': '%d synthetic 2x150 bp reads per GPU, %d read group(s), Q0-41, '
                                   'accumulate + solve + apply end-to-end, device-resident' % (n, R),
                       'reads_per_gpu': n, 'read_len': S, 'read_groups': R, 'layout': layout_name,
                       'layout_written_by': 'in the product: the FASTQ packer itself (kbbq_fastq_fill_rows fills these rows into the '
                                            'page-locked upload slabs, no device pass); here: synthetic reads generated on the device as '
                                            'character rows and laid out before the timed region by the pass that writes the same bytes '
                                            '(tests/test_gpu_layouts.py::test_the_packer_writes_the_rows_the_layout_pass_writes)',
                       'layout_pass_ms_before_the_timed_region': layout_pass_ms,
                       'layout_inclusive_bases_per_s': (None if layout_pass_ms is None else
                                                        bases_per_rank * world / ((step_ms + layout_pass_ms) * 1e-3)),
                       'parallelism': 'reads sharded x%d, 1 allreduce of count tables' % world},
            'verified': verified, 'verified_per_rank': verified_ranks,
            'verified_how': 'after the timed region, on EVERY rank (the line carries the AND): the new qualities of the first 1 M reads of '
                            "the rank's resident batch (in the headline's layout, as the last step left them) == the persistent apply "
                            'kernel on the same reads as character rows, one read per row',
            'host_binding': host_binding,
            'ranks_seen': ranks_seen, 'backend': backend,
            'allreduce_ms_per_step': ar_ms,
            'per_rank_ms_per_step': {'min': min(per_rank_ms), 'max': max(per_rank_ms), 'ranks': per_rank_ms},
            'roofline': {'bound': 'hbm', 'kernel': dom, 'achieved': domk['GB/s'], 'peak': HBM_PEAK_GBS,
                         'unit': 'GB/s', 'frac': domk['frac'],
                         'bytes_per_base_algorithmic': 3,
                         'bytes_per_base_of_the_layout': layout_bytes[dom], 'achieved_on_layout_bytes': gbs(layout_bytes[dom] * bases_per_rank, domk['avg_ms']),
                         'frac_on_layout_bytes': gbs(layout_bytes[dom] * bases_per_rank, domk['avg_ms']) / HBM_PEAK_GBS,
                         'traffic': None if per_base is None else per_base * bases_per_rank,
                         'traffic_source': 'profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE x 2 + WRITE_SIZE, separate '
                                           'passes, 20 M reads, scaled per base; null when taken on another kernel source)',
                         'kernel_source_sha': kernel_source_sha()},
            'kernels': {'k1_accumulate': k1, 'k2_apply': k2,
                        'host_solve_and_sync_ms': step_ms - k1_avg - k2_avg},
        }
        if cpu_base is not None:
            base = cpu_base
            ref = reference_figure()
            if ref is not None:
                base['reference_bases_per_s'] = ref.get('end_to_end_bases_per_s')
                base['reference_pass1_bases_per_s'] = ref.get('pass1_bases_per_s')
                base['reference_cores'] = ref.get('cores_used', 1)
                base['reference_note'] = ('the unmodified reference (Python / NumPy, single-threaded) on BASELINE config 1, measured in '
                                          'the build container (%s host cores there) by oracle/gen_golden.py: tests/golden/c1_10k_1rg.json'
                                          % ref.get('host_cores', '8'))
            base['host_cores'] = host_cores()
            base['measured'] = 'before the first GPU call of this process (the worker pool of the all-core leg is forked from a process without a HIP context)'
            res_line['cpu_baseline'] = base
        if world == 1 and not args.no_extra:
            res_line['extra'] = build_extra(torch, dev, parallel, args, args.layout)
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(res_line), flush=True)
        os.dup2(2, 1)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main(argv=None):
    args = parse_args(argv)
    if args.gpus > 1 and 'RANK' not in os.environ:
        return launch_children(args)
    return run_rank(args)


if __name__ == '__main__':
    sys.exit(main())
