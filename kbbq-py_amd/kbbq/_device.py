"""
Device plumbing for the hot path: contexts, device-resident read batches and
count tables.  PyTorch provides device memory, streams and (in parallel.py) the
RCCL process group; every computation is a libkbbq_hip kernel reached through
the C ABI (kbbq._native).  No CPU fallback exists.
"""
import ctypes
import os

import numpy as np

from . import _native as N

MINSCORE = 6
MAXSCORE = 42
NQ = MAXSCORE + 1

_contexts = {}


_backend = None          # the module that provides device memory, streams and events: torch, or kbbq._hipmem (use_native_memory)


def use_native_memory():
    """Device memory, page-locked buffers, copies and events from libkbbq_hip's own C ABI (kbbq/_hipmem.py) instead of
    PyTorch, for the rest of the process: the single-GPU command line calls this before it touches the device and never
    imports torch (its import, HIP context and kernels' code objects were ~1 s of a 2.4 s command).  Refused once torch
    tensors are in use (the two kinds of buffers do not mix); multi-GPU runs keep torch for the RCCL process group."""
    global _backend
    from . import _hipmem
    if _backend is not None and _backend is not _hipmem:
        raise RuntimeError('use_native_memory(): this process already uses torch for device memory')
    if not _hipmem.cuda.is_available():
        raise N.KbbqHipError('no HIP device available: kbbq needs an MI355X (gfx950); there is no CPU fallback')
    _backend = _hipmem
    return _hipmem


def _torch():
    global _backend
    if _backend is None:
        import torch
        if not torch.cuda.is_available():
            raise N.KbbqHipError('torch sees no GPU: kbbq needs an MI355X (gfx950); there is no CPU fallback')
        _backend = torch
    return _backend


def parse_bytes(text):
    """'256M', '1.5G', '64k', '1000000' -> bytes."""
    t = str(text).strip()
    mult = {'k': 1 << 10, 'm': 1 << 20, 'g': 1 << 30, 't': 1 << 40}.get(t[-1:].lower())
    if t[-2:].lower() in ('kb', 'mb', 'gb', 'tb'):
        t, mult = t[:-1], {'k': 1 << 10, 'm': 1 << 20, 'g': 1 << 30, 't': 1 << 40}[t[-2].lower()]
    return int(float(t[:-1]) * mult) if mult else int(float(t))


def device_budget():
    """Device memory the file path may hold for reads at any one time: KBBQ_DEVICE_BUDGET (bytes, or with a K / M / G
    suffix), else 60 % of what is free on the device now.  A shard whose resident form needs more is streamed
    (kbbq/_stream.py): slabs of rows sized to this budget go through K1 (pass 1) and K2 (pass 2) and are dropped."""
    env = os.environ.get('KBBQ_DEVICE_BUDGET')
    if env:
        return max(parse_bytes(env), 1 << 20)
    torch = _torch()
    free, _ = torch.cuda.mem_get_info()
    return int(free * 0.6)


def memory_peak(reset=False):
    """Largest number of bytes of device memory in use by tensors since the last reset (torch's allocator, or
    kbbq/_hipmem.py's accounting of kbbq_dev_alloc)."""
    torch = _torch()
    if reset:
        torch.cuda.reset_peak_memory_stats()
        return 0
    return int(torch.cuda.max_memory_allocated())


def synchronize():
    """Wait for everything enqueued on the current device (stage timing)."""
    if _backend is not None:
        _backend.cuda.synchronize()


def warm_up(device=0):
    """Everything the first call of a process pays once -- importing torch (unless kbbq._hipmem stands in), creating the HIP context, loading the code objects of libkbbq_hip and of the torch kernels the
    path uses -- as a miniature run (64 synthetic pairs: tally, solve, apply), so that a caller can pay it while the
    host reads its input files.  Does nothing the second time."""
    global _warm
    if _warm:
        return
    torch = _torch()
    with torch.cuda.device(device):
        batch = ReadBatch.synthetic(0, 128, 128, seed=1)
        tables = Tables(1, 2 * 150)
        accumulate(batch, tables)
        if torch.__name__ != 'torch' and device not in _logtabs and not os.environ.get('KBBQ_HOST_SOLVE'):
            # the command line without torch: the proof of the device's gammaln against the host's (1.3 M arguments, ~40 ms) is not
            # needed before the first solve -- it runs on a thread of its own while the packer fills and uploads the reads
            # (device_logtab() joins it); the first launches above have loaded the library's code object
            import threading
            job = threading.Thread(target=_prove_logtab, args=(device,), daemon=True)
            _logtab_jobs[device] = job
            job.start()
            torch.cuda.synchronize()
        else:
            lut, shape = solve_lut(tables)        # also proves (once per process) the device's gammaln against the host's
            apply(batch, lut, shape).cpu()
            torch.cuda.synchronize()
    _warm = True


_warm = False


def context(device=None):
    """The process-wide kbbq context of a device, bound to torch's current stream."""
    torch = _torch()
    if device is None:
        device = torch.cuda.current_device()
    ctx = _contexts.get(device)
    if ctx is None:
        ctx = N.Context(device)
        _contexts[device] = ctx
    if torch.__name__ == 'torch':            # without torch the context keeps its own stream (kbbq/_hipmem.py enqueues there too)
        ctx.set_stream(torch.cuda.current_stream(device).cuda_stream)
    return ctx


class Tables:
    """The four count arrays of recalibrate.py:51-54 as ONE int64 device buffer
    [pos_errs | pos_total | dinuc_errs | dinuc_total] (see include/kbbq_hip.h)."""

    def __init__(self, R, S2, device=None):
        torch = _torch()
        self.R, self.S2 = int(R), int(S2)
        n = N.load().kbbq_tables_count(self.R, self.S2)
        self.buf = torch.zeros(n, dtype=torch.int64, device='cuda' if device is None else device)

    @classmethod
    def from_host(cls, pos_errs, pos_total, dinuc_errs, dinuc_total, device=None):
        """Upload stored count arrays (a model read back from a GATK report)."""
        torch = _torch()
        pos_errs = np.asarray(pos_errs)
        R, nq, S2 = pos_errs.shape
        if nq != NQ or np.shape(pos_total) != (R, NQ, S2) or np.shape(dinuc_errs) != (R, NQ, 16) \
                or np.shape(dinuc_total) != (R, NQ, 16):
            raise ValueError('count arrays must be [R,43,2S], [R,43,2S], [R,43,16], [R,43,16]')
        t = cls(R, S2, device=device)
        flat = np.concatenate([np.ascontiguousarray(a, dtype=np.int64).ravel()
                               for a in (pos_errs, pos_total, dinuc_errs, dinuc_total)])
        t.buf.copy_(torch.from_numpy(flat))
        return t

    def add(self, other):
        """self += other (same shape), by the library's own kernel on the launch stream."""
        if (other.R, other.S2) != (self.R, self.S2):
            raise ValueError('count tables of different shapes')
        ctx = context(self.buf.device.index)
        N.check(N.load().kbbq_tables_add_dev(ctx.handle, N.ptr(self.buf), N.ptr(other.buf), self.buf.numel()))

    def views(self):
        R, S2 = self.R, self.S2
        npos, ndn = R * NQ * S2, R * NQ * 16
        b = self.buf
        return (b[:npos].view(R, NQ, S2), b[npos:2 * npos].view(R, NQ, S2),
                b[2 * npos:2 * npos + ndn].view(R, NQ, 16), b[2 * npos + ndn:].view(R, NQ, 16))

    def host_views(self):
        """(pos_errs, pos_total, dinuc_errs, dinuc_total) as views of ONE host copy of the buffer (read-only use)."""
        h = self.buf.cpu().numpy()
        R, S2 = self.R, self.S2
        npos, ndn = R * NQ * S2, R * NQ * 16
        return (h[:npos].reshape(R, NQ, S2), h[npos:2 * npos].reshape(R, NQ, S2),
                h[2 * npos:2 * npos + ndn].reshape(R, NQ, 16), h[2 * npos + ndn:].reshape(R, NQ, 16))

    def to_host(self):
        """(pos_errs, pos_total, dinuc_errs, dinuc_total) as fresh int64 numpy arrays."""
        return tuple(a.copy() for a in self.host_views())


_pinned = {}
_pinned_lock = __import__('threading').RLock()      # the streaming path's producer and the egress pipeline's copy stage both ask for buffers


def pinned(tag, index, nbytes):
    """A page-locked host buffer (uint8 tensor of at least nbytes) kept for the life of the process: page-locking is
    slow (milliseconds per MB), so the staging buffers of the file path -- ingest slabs, then the egress pipeline's --
    are allocated once and shared by index; `tag` separates users whose buffers are in use at the same time."""
    torch = _torch()
    key = (tag, index)
    with _pinned_lock:
        buf = _pinned.get(key)
        if buf is None or buf.shape[0] < nbytes:
            for other, cand in list(_pinned.items()):                # a free buffer of another tag that is large enough
                if other[0] == '' and cand.shape[0] >= nbytes:
                    buf = _pinned.pop(other)
                    break
            else:
                buf = torch.empty(int(nbytes), dtype=torch.uint8, pin_memory=True)
            _pinned[key] = buf
    return buf


_released = [0]


def release_pinned(tag):
    """The buffers of `tag` become available to other users (kept allocated)."""
    with _pinned_lock:
        for key in [k for k in _pinned if k[0] == tag]:
            _released[0] += 1
            _pinned[('', _released[0])] = _pinned.pop(key)


class ReadBatch:
    """Device-resident padded SoA of reads: seq / cseq / qual planes [n, pitch] uint8
    and the uint32 sidecar (layout: include/kbbq_hip.h)."""

    nib = False          # seq / cseq hold one code nibble per base (include/kbbq_hip.h KBBQ_ROWS_NIBBLES)
    seg = None           # rows grouped by read group: int64 [R + 1] on the device
    perm = None          # ... and row i of this batch is row perm[i] of the batch it was made from

    def __init__(self, n, pitch, with_corrected=True, device=None, nib=False):
        torch = _torch()
        dev = 'cuda' if device is None else device
        self.n, self.pitch, self.nib = int(n), int(pitch), bool(nib)
        rows = max(self.n, 1)
        sp = pitch // 2 if nib else pitch
        self.seq = torch.empty((rows, sp), dtype=torch.uint8, device=dev)
        self.qual = torch.empty((rows, pitch), dtype=torch.uint8, device=dev)
        self.cseq = torch.empty((rows, sp), dtype=torch.uint8, device=dev) if with_corrected else None
        self.meta = torch.empty(rows, dtype=torch.int32, device=dev)

    def describe(self):
        return _describe(self, 'one read per row, pitch %d' % self.pitch)

    def layout_key(self):
        return 'reads_nib' if self.nib else 'reads'

    def chars(self, name='seq'):
        """The seq / cseq plane as characters [n, pitch] whatever the layout stores (tests, debugging)."""
        return _plane_chars(self, name)

    @classmethod
    def from_host(cls, seq, qual, meta, cseq=None, device=None):
        torch = _torch()
        n, pitch = seq.shape
        b = cls(n, pitch, with_corrected=cseq is not None, device=device)
        if n:
            b.seq[:n].copy_(torch.from_numpy(np.ascontiguousarray(seq)))
            b.qual[:n].copy_(torch.from_numpy(np.ascontiguousarray(qual)))
            if cseq is not None:
                b.cseq[:n].copy_(torch.from_numpy(np.ascontiguousarray(cseq)))
            b.meta[:n].copy_(torch.from_numpy(np.ascontiguousarray(meta).view(np.int32)))
        return b

    @classmethod
    def from_reader(cls, text, other, infer_rg_flag, first, n, pitch, slab=1 << 17, device=None, keep_pinned=False):
        """Reads [first, first + n) of a fastx.NativeFastq (and their corrections from `other`, or None) straight onto
        the device: the C++ packer fills page-locked slabs (all host threads) while the copy engine uploads the
        previous ones -- no host copy of the planes is ever materialised."""
        torch = _torch()
        b = cls(n, pitch, with_corrected=other is not None, device=device)
        planes = 3 if other is not None else 2
        events = {}
        with torch.cuda.device(b.seq.device):
            for k, lo in enumerate(range(0, n, slab)):
                m = min(slab, n - lo)
                slot = k % 3
                if slot in events:
                    events[slot].synchronize()                   # the slab's previous upload has left the buffer
                buf = pinned('ingest', slot, slab * (planes * pitch + 4))
                view = lambda i: buf[i * slab * pitch:i * slab * pitch + m * pitch].view(m, pitch)
                h_seq, h_qual = view(0), view(1)
                h_cseq = view(2) if other is not None else None
                h_meta = buf[planes * slab * pitch:planes * slab * pitch + 4 * m].view(torch.int32)
                text.fill_into(other, infer_rg_flag, m, pitch, first + lo, h_seq.numpy(), None if h_cseq is None else h_cseq.numpy(),
                               h_qual.numpy(), h_meta.numpy().view(np.uint32))
                b.seq[lo:lo + m].copy_(h_seq, non_blocking=True)
                b.qual[lo:lo + m].copy_(h_qual, non_blocking=True)
                if other is not None:
                    b.cseq[lo:lo + m].copy_(h_cseq, non_blocking=True)
                b.meta[lo:lo + m].copy_(h_meta, non_blocking=True)
                events[slot] = torch.cuda.Event()
                events[slot].record()
            for e in events.values():
                e.synchronize()
        if not keep_pinned:                  # (the streaming path fills slab after slab: it releases them once, at the end)
            release_pinned('ingest')
        b.h2d_bytes = n * (planes * pitch + 4)
        return b

    @classmethod
    def synthetic(cls, first, n, total, seed, len_lo=150, len_hi=150, nrg=1, qlo=0, qhi=41,
                  pitch=None, device=None):
        """Synthetic reads generated on the device (spec in oracle/kbbq_oracle.c)."""
        from ._synth import SYNTH_THR
        if pitch is None:
            pitch = max(16, (len_hi + 15) // 16 * 16)
        b = cls(n, pitch, with_corrected=True, device=device)
        ctx = context(b.seq.device.index)
        N.check(N.load().kbbq_synth_dev(ctx.handle, N.ptr(b.seq), N.ptr(b.cseq), N.ptr(b.qual),
                                        N.ptr(b.meta), first, n, total, pitch, seed,
                                        len_lo, len_hi, nrg, qlo, qhi, N.ptr(SYNTH_THR)))
        return b

    def lengths_host(self):
        return (self.meta[:self.n].cpu().numpy().view(np.uint32) & 0xFFFF).astype(np.int64)


class PairBatch:
    """Device-resident mate-pair rows (include/kbbq_hip.h "mate-pair rows"): one row per read pair,
    [mate 1][separator][mate 2][padding], pitch = roundup16(2S + 1) -- 304 instead of 2 x 160 bytes
    for 2 x 150 bp.  accumulate() / apply() take it like a ReadBatch and give identical results;
    `n` counts rows (pairs)."""

    nib = False
    seg = None
    perm = None
    twins = False        # both reads of a row are first in pair: single-end input packed two to a row (KBBQ_ROWS_TWINS)

    def __init__(self, npairs, S, with_corrected=True, device=None, nib=False):
        torch = _torch()
        dev = 'cuda' if device is None else device
        self.n, self.S, self.nib = int(npairs), int(S), bool(nib)
        self.pitch = int(N.load().kbbq_pair_pitch(2 * self.S))
        rows = max(self.n, 1)
        sp = self.pitch // 2 if nib else self.pitch
        self.seq = torch.empty((rows, sp), dtype=torch.uint8, device=dev)
        self.qual = torch.empty((rows, self.pitch), dtype=torch.uint8, device=dev)
        self.cseq = torch.empty((rows, sp), dtype=torch.uint8, device=dev) if with_corrected else None
        self.meta = torch.empty(rows, dtype=torch.int32, device=dev)
        self.read_pitch = None

    def describe(self):
        return _describe(self, ('two single-end reads per row, pitch %d' if self.twins else 'mate-pair rows, pitch %d per pair') % self.pitch)

    def layout_key(self):
        return 'pairs_nib' if self.nib else 'pairs'

    def chars(self, name='seq'):
        return _plane_chars(self, name)

    @staticmethod
    def worthwhile(S, pitch):
        """True when pair rows move fewer bytes than two rows of `pitch`."""
        return int(N.load().kbbq_pair_pitch(2 * int(S))) < 2 * int(pitch)

    @classmethod
    def from_reads(cls, batch):
        """Re-lay a ReadBatch whose reads alternate first / second in pair, all of one length, mates in
        one read group.  ValueError otherwise (use the ReadBatch as it is)."""
        n = batch.n
        if n == 0 or n % 2:
            raise ValueError('mate-pair rows need an even, non-zero number of reads')
        if batch.nib:
            raise ValueError('mate-pair rows are made from character planes (use lay_out)')
        st = meta_stats(batch)
        S = st['longest']
        if st['pair_violations'] or S == 0:
            raise ValueError('reads are not uniform first/second pairs of one length and read group')
        pb = cls(n // 2, S, with_corrected=batch.cseq is not None, device=batch.seq.device)
        pb.read_pitch = batch.pitch
        ctx = context(batch.seq.device.index)
        N.check(N.load().kbbq_pack_pairs_dev(ctx.handle, N.ptr(batch.seq), N.ptr(batch.cseq), N.ptr(batch.qual),
                                             N.ptr(batch.meta), pb.n, batch.pitch, 2 * S,
                                             N.ptr(pb.seq), N.ptr(pb.cseq), N.ptr(pb.qual), N.ptr(pb.meta)))
        return pb

    def unpack(self, plane, pitch=None):
        """A pair-row plane (the apply output) as one-read-per-row [2 n, pitch]."""
        torch = _torch()
        pitch = int(pitch or self.read_pitch or (self.S + 15) // 16 * 16)
        out = torch.empty((max(2 * self.n, 1), pitch), dtype=torch.uint8, device=plane.device)
        ctx = context(plane.device.index)
        N.check(N.load().kbbq_unpack_pairs_dev(ctx.handle, N.ptr(plane), self.n, 2 * self.S, pitch, N.ptr(out)))
        return out


def meta_stats(batch):
    """One device pass over the sidecar words of a ReadBatch (k7_meta_stats): shortest non-empty / longest read, largest
    read-group id, violations of the mate-pair preconditions, empty reads."""
    ctx = context(batch.meta.device.index)
    st = np.zeros(8, dtype=np.int32)
    N.check(N.load().kbbq_meta_stats_dev(ctx.handle, N.ptr(batch.meta), batch.n, N.ptr(st)))
    shortest = int(st[0]) if st[0] != 0x7FFFFFFF else 0
    return {'shortest': shortest, 'longest': int(st[1]), 'max_rg': int(st[2]), 'pair_violations': int(st[3]), 'empty': int(st[4]),
            'twin_violations': int(st[5])}


def _describe(batch, base):
    if batch.nib:
        base += ', 4-bit sequence planes'
    if batch.seg is not None:
        base += ', rows grouped by read group'
    return base


def _plane_chars(batch, name):
    torch = _torch()
    plane = getattr(batch, name)
    if not batch.nib:
        return plane
    out = torch.empty((plane.shape[0], batch.pitch), dtype=torch.uint8, device=plane.device)
    ctx = context(plane.device.index)
    N.check(N.load().kbbq_unpack_nibbles_dev(ctx.handle, N.ptr(plane), plane.shape[0] * batch.pitch, N.ptr(out)))
    return out


GROUP_NATIVE_MAX_R = 256


def _group_perm(meta, nrows, pairs, R):
    """(perm, seg) of the stable sort of `nrows` rows by read group (pairs: row = pair, its group the first mate's):
    the library's counting sort (kbbq_group_rows_dev) up to 256 groups, a torch sort beyond."""
    torch = _torch()
    dev_ = meta.device
    if R <= GROUP_NATIVE_MAX_R:
        lib = N.load()
        ctx = context(dev_.index)
        work = torch.empty(lib.kbbq_group_rows_work_bytes(nrows, R), dtype=torch.uint8, device=dev_)
        perm = torch.empty(max(nrows, 1), dtype=torch.int64, device=dev_)
        seg = torch.empty(R + 1, dtype=torch.int64, device=dev_)
        N.check(lib.kbbq_group_rows_dev(ctx.handle, N.ptr(meta), nrows, 1 if pairs else 0, R, N.ptr(work), N.ptr(perm), N.ptr(seg)))
        try:
            ctx.status()
        except ValueError as e:
            raise ValueError('a row carries a read group >= R = %d (%s)' % (R, e)) from None
        return perm[:nrows], seg
    rg = ((meta[:2 * nrows:2] if pairs else meta[:nrows]) >> 16) & 0x7FFF
    if nrows and int(rg.max().item()) >= R:
        raise ValueError('a row carries read group %d but R = %d' % (int(rg.max().item()), R))
    perm = torch.argsort(rg, stable=True)
    seg = torch.zeros(R + 1, dtype=torch.int64, device=dev_)
    seg[1:] = torch.cumsum(torch.bincount(rg, minlength=R), 0)
    return perm, seg


def _lay_out_rows(batch, flags, S2, perm, seg):
    """kbbq_lay_out_dev: input-order rows -> the destination layout, one pass."""
    pairs, nib = bool(flags & N.ROWS_PAIRS), bool(flags & N.ROWS_NIBBLES)
    if pairs:
        laid = PairBatch((batch.n + 1) // 2, S2 // 2, with_corrected=batch.cseq is not None, device=batch.seq.device, nib=nib)
        laid.read_pitch, laid.twins = batch.pitch, bool(flags & N.ROWS_TWINS)
    elif isinstance(batch, PairBatch):                # pair rows gathered as rows: still pair rows
        laid = PairBatch(batch.n, batch.S, with_corrected=batch.cseq is not None, device=batch.seq.device, nib=nib)
        laid.read_pitch, laid.twins = batch.read_pitch, batch.twins
        flags &= ~N.ROWS_TWINS
    else:
        laid = ReadBatch(batch.n, batch.pitch, with_corrected=batch.cseq is not None, device=batch.seq.device, nib=nib)
    ctx = context(batch.seq.device.index)
    N.check(N.load().kbbq_lay_out_dev(ctx.handle, N.ptr(batch.seq), N.ptr(batch.cseq), N.ptr(batch.qual), N.ptr(batch.meta),
                                      batch.n, batch.pitch, flags, S2, N.ptr(perm),
                                      N.ptr(laid.seq), N.ptr(laid.cseq), N.ptr(laid.qual), N.ptr(laid.meta)))
    laid.seg, laid.perm = seg, perm
    return laid


def layout_flags(st, n, pitch, packed=True, pairs=None):
    """The KBBQ_ROWS_* bits of the layout `n` reads qualify for, from their sidecar statistics `st` (meta_stats on the
    device, fastx.NativeFastq.meta on the host -- the same numbers): two reads to a row for uniform first / second mates
    -- or single-end neighbours (`twins`; an odd number is fine: the last row's second half stays padding) -- of one
    length and read group when such rows are narrower than two rows of `pitch`; 4-bit planes when `packed`.  pairs=True
    insists on two reads to a row (ValueError if the reads do not qualify), pairs=False forbids it."""
    S_ = st['longest']
    fits = n % 2 == 0 and S_ > 0
    twin_fits = S_ > 0 and n >= 2 and st['twin_violations'] == 0
    twins = False
    if pairs is None:
        pairs = fits and st['pair_violations'] == 0 and PairBatch.worthwhile(S_, pitch)
        if not pairs and twin_fits and PairBatch.worthwhile(S_, pitch):
            pairs = twins = True
    elif pairs and not (fits and st['pair_violations'] == 0):
        if not twin_fits:
            raise ValueError('reads are not uniform first/second pairs (or single-end neighbours) of one length and read group')
        twins = True
    return (N.ROWS_PAIRS if pairs else 0) | (N.ROWS_NIBBLES if packed else 0) | (N.ROWS_TWINS if twins else 0)


def lay_out(batch, R, S=None, packed=True, pairs=None, stats=None):
    """The device layout K1 / K2 run fastest on for this input-order ReadBatch, written in ONE pass
    (k7_lay_out): mate-pair rows when the reads are uniform first / second pairs of one length and read group (and
    pair rows are narrower than two rows) -- or single-end reads of one length whose neighbours share a read group: the
    same rows with `twins` set, both halves counting into the forward cycle columns --, rows gathered by read-group segment when R > 1, 4-bit sequence planes when
    `packed` and every base of seq AND cseq is one of ACGTN (otherwise the pass is repeated with character planes).
    The result carries `perm` (apply(..., restore_order=True) stores straight back into input order) and `seg`.
    Returns `batch` itself when no layout applies."""
    if batch.nib or batch.seg is not None or isinstance(batch, PairBatch):
        raise ValueError('lay_out takes input-order rows of characters')
    if batch.n == 0:
        return batch
    st = stats or meta_stats(batch)
    S_ = st['longest']
    flags = layout_flags(st, batch.n, batch.pitch, packed, pairs)
    pairs = bool(flags & N.ROWS_PAIRS)
    perm = seg = None
    if R > 1:
        perm, seg = _group_perm(batch.meta, (batch.n + 1) // 2 if pairs else batch.n, pairs, R)
    if not flags and perm is None:
        return batch
    ctx = context(batch.seq.device.index)
    laid = _lay_out_rows(batch, flags, 2 * S_, perm, seg)
    if packed:
        try:
            ctx.status()
        except N.LutNeedsCheckedApply:                # a base outside ACGTN: character planes keep the exact semantics
            if not (flags & ~N.ROWS_NIBBLES) and perm is None:
                return batch
            laid = _lay_out_rows(batch, flags & ~N.ROWS_NIBBLES, 2 * S_, perm, seg)
    return laid


def laid_from_reader(text, other, infer_rg_flag, first, n, pitch, R, packed=True, pairs=None, slab=1 << 17, device=None, pair_S=None,
                     keep_pinned=False, with_out=False):
    """Reads [first, first + n) of a fastx.NativeFastq (and their corrections from `other`, or None) onto the device IN
    THE LAYOUT lay_out() would give them -- written by the C++ packer itself (kbbq_fastq_fill_rows): mate-pair rows, 4-bit
    sequence planes, rows gathered by read-group segment, decided from the sidecar statistics of the text (kbbq_fastq_meta)
    before a byte is uploaded.  No character rows ever exist, on the host or on the device, and no layout pass runs:
    2 B/base cross PCIe instead of 3.  Page-locked slabs are filled by all host threads while the copy engine uploads the
    previous ones, as in ReadBatch.from_reader.  Returns None when no layout applies (plain rows: ReadBatch.from_reader);
    a letter outside ACGTN repeats the fill with character planes (the reference's TypeError rule lives there)."""
    torch = _torch()
    if n == 0:
        return None
    meta, st = text.meta(infer_rg_flag, first, n)
    S_ = st['longest']
    if pair_S is not None and S_ != pair_S and pairs is None:
        pairs = False                        # two reads to a row need count tables / a LUT of exactly 2 x THESE reads' length (a slab of a band)
    flags = layout_flags(st, n, pitch, packed, pairs)
    two = bool(flags & N.ROWS_PAIRS)
    nrows = (n + 1) // 2 if two else n
    perm = seg = None
    if R > 1:
        if st['max_rg'] >= R:
            raise ValueError('a row carries read group %d but R = %d' % (st['max_rg'], R))
        perm, seg = np.empty(nrows, dtype=np.int64), np.empty(R + 1, dtype=np.int64)
        N.check(N.load().kbbq_group_rows_host(N.ptr(meta), nrows, 1 if two else 0, R, N.ptr(perm), N.ptr(seg)))
    while True:
        if not flags and perm is None:
            return None
        nib = bool(flags & N.ROWS_NIBBLES)
        if two:
            laid = PairBatch(nrows, S_, with_corrected=other is not None, device=device, nib=nib)
            laid.read_pitch, laid.twins = pitch, bool(flags & N.ROWS_TWINS)
        else:
            laid = ReadBatch(n, pitch, with_corrected=other is not None, device=device, nib=nib)
        if with_out:                         # the output plane's first-use wipe passes behind the fill (_take_out_plane)
            laid.out_plane = torch.empty_like(laid.qual)
        dp = laid.pitch
        sp = dp // 2 if nib else dp
        nseq = 2 if other is not None else 1
        step = max(slab // 2 if two else slab, 1)                 # destination rows per slab: the same bytes either way
        row_bytes = nseq * sp + dp + 4
        events, foreign = {}, False
        with torch.cuda.device(laid.seq.device):
            for k, lo in enumerate(range(0, nrows, step)):
                m = min(step, nrows - lo)
                slot = k % 3
                if slot in events:
                    events[slot].synchronize()                    # the slab's previous upload has left the buffer
                buf = pinned('ingest', slot, step * row_bytes)
                at = [0]

                def view(width, rows=m, cap=step):
                    v = buf[at[0]:at[0] + rows * width].view(rows, width)
                    at[0] += cap * width
                    return v
                h_seq = view(sp)
                h_cseq = view(sp) if other is not None else None
                h_qual = view(dp)
                h_meta = buf[at[0]:at[0] + 4 * m].view(torch.int32)
                foreign |= text.fill_rows(other, first, n, meta, flags, 2 * S_, dp, perm, lo, m, h_seq.numpy(),
                                          None if h_cseq is None else h_cseq.numpy(), h_qual.numpy(), h_meta.numpy().view(np.uint32))
                if foreign:
                    break
                laid.seq[lo:lo + m].copy_(h_seq, non_blocking=True)
                laid.qual[lo:lo + m].copy_(h_qual, non_blocking=True)
                if other is not None:
                    laid.cseq[lo:lo + m].copy_(h_cseq, non_blocking=True)
                laid.meta[lo:lo + m].copy_(h_meta, non_blocking=True)
                events[slot] = torch.cuda.Event()
                events[slot].record()
            for e in events.values():
                e.synchronize()
        if not keep_pinned:
            release_pinned('ingest')
        if not foreign:
            break
        flags &= ~N.ROWS_NIBBLES                                  # a base outside ACGTN: character planes keep the exact semantics
    laid.h2d_bytes = nrows * row_bytes                            # what crossed PCIe for this band (the file path's trace reports it)
    if perm is not None:
        laid.perm = torch.from_numpy(perm).to(laid.seq.device)
        laid.seg = torch.from_numpy(seg).to(laid.seq.device)
        laid.h2d_bytes += perm.nbytes + seg.nbytes
    return laid


def group_by_rg(batch, R):
    """The same rows ordered by read group (stable), with `seg` (int64 [R + 1] on the device: group g owns rows
    [seg[g], seg[g + 1])) and `perm` (row i of the result is row perm[i] of `batch`).  accumulate() / apply()
    then run every group at the single-group rate (include/kbbq_hip.h "rows grouped by read group");
    apply(..., restore_order=True) stores straight back into the original order (ungroup() does it as a separate
    pass).  Works on ReadBatch and PairBatch (character planes)."""
    import copy
    if batch.nib:
        raise ValueError('group_by_rg takes character planes (lay_out groups and packs in one pass)')
    n = batch.n
    perm, seg = _group_perm(batch.meta, n, False, R)
    if n == 0:
        g = copy.copy(batch)
        g.seg, g.perm = seg, perm
        return g
    return _lay_out_rows(batch, 0, 2 * getattr(batch, 'S', 0), perm, seg)


def ungroup(batch, plane):
    """An output plane of a grouped batch, rows back in the order before group_by_rg."""
    torch = _torch()
    n = batch.n
    out = torch.empty_like(plane[:max(n, 1)])
    if n:
        out.index_copy_(0, batch.perm, plane[:n])
    return out


_pair_luts = {}


def _row_flags(batch):
    pairs = isinstance(batch, PairBatch)
    return (N.ROWS_PAIRS if pairs else 0) | (N.ROWS_NIBBLES if batch.nib else 0) | (N.ROWS_TWINS if pairs and batch.twins else 0)


LONG_READS = 160         # beyond this a band's shortest read decides whether K1's LDS tables fit (kbbq_accumulate_band_dev)
PACKED_READS = int(os.environ.get('KBBQ_PACKED_READS', '320'))      # longest read the file path still lays out on 4-bit planes


def accumulate(batch, tables, minscore=MINSCORE, check=True, dinuc_minscore=None, s_band=0, s_min=0):
    """K1 over a device batch, adding into `tables` (recalibrate.py:57-119).  s_band: the longest read of THIS batch
    when it is one length band of a mixed-length input (its rows packed at a narrower pitch than the tables' S):
    the kernel's LDS tables are then laid out for s_band instead of tables.S2 / 2.  s_min: no non-empty read of the
    batch is shorter; it lets bands of ~200-300-base reads fit the table-driven kernel.  Left at 0 for such reads,
    both are measured on the device (k7_meta_stats) before the launch."""
    ctx = context(batch.seq.device.index)
    pairs = isinstance(batch, PairBatch)
    if pairs and tables.S2 != 2 * batch.S:
        raise ValueError('mate-pair rows of %d-base reads need tables with 2S = %d columns' % (batch.S, 2 * batch.S))
    if not pairs and not s_min and batch.n and (s_band or tables.S2 // 2) > LONG_READS:
        st = meta_stats(batch)
        s_min = st['shortest']
        s_band = s_band or min(max(st['longest'], 1), tables.S2 // 2)
    dm = minscore if dinuc_minscore is None else dinuc_minscore
    if batch.seg is not None or batch.nib or pairs:
        N.check(N.load().kbbq_accumulate_rows_dev(ctx.handle, N.ptr(batch.seq), N.ptr(batch.cseq), N.ptr(batch.qual),
                                                  N.ptr(batch.meta), batch.n, batch.pitch, _row_flags(batch),
                                                  tables.R, tables.S2, int(s_band), int(s_min), minscore, dm,
                                                  N.ptr(batch.seg), N.ptr(tables.buf)))
    else:
        N.check(N.load().kbbq_accumulate_band_dev(ctx.handle, N.ptr(batch.seq), N.ptr(batch.cseq),
                                                  N.ptr(batch.qual), N.ptr(batch.meta), batch.n, batch.pitch,
                                                  tables.R, tables.S2, int(s_band), int(s_min), minscore, dm,
                                                  N.ptr(tables.buf)))
    if check:
        ctx.status()


def _addr(x):
    return None if x is None else x.data_ptr()


def _band_array(items, outs=None, pluts=None, restore_order=False):
    """ctypes array of kbbq_band for `items` = [(batch, s_band, s_min)]."""
    arr = (N.Band * max(len(items), 1))()
    for k, (batch, s_band, s_min) in enumerate(items):
        pairs = isinstance(batch, PairBatch)
        b = arr[k]
        b.d_seq, b.d_cseq, b.d_qual, b.d_meta = _addr(batch.seq), _addr(batch.cseq), _addr(batch.qual), _addr(batch.meta)
        b.nrows, b.pitch, b.flags = batch.n, batch.pitch, _row_flags(batch)
        b.S_band, b.S_min = (0, 0) if pairs else (int(s_band), int(s_min))
        b.d_seg = _addr(batch.seg)
        b.d_perm = _addr(batch.perm) if (restore_order and batch.seg is not None) else None
        b.d_out = None if outs is None else _addr(outs[k])
        b.d_pair_lut = None if pluts is None else _addr(pluts[k])
    return arr


def _take_out_plane(batch):
    """The plane K2 writes a batch's new qualities into: the one the file path allocated beside the input planes while the packer
    was still filling them (laid_from_reader: device memory is wiped by the driver before its first use -- 1.2 GB at ~30 GB/s is
    the 40 ms the command line's first apply used to wait -- and that wait now passes behind the host's fill), else a fresh one."""
    torch = _torch()
    out = getattr(batch, 'out_plane', None)
    if out is not None:
        batch.out_plane = None               # handed over once: the egress pipeline releases it slab by slab
        return out
    return torch.empty_like(batch.qual)


def accumulate_bands(items, tables, minscore=MINSCORE, check=True, dinuc_minscore=None):
    """K1 over ALL length bands of a mixed-length input in one launch (kbbq_accumulate_bands_dev: every band on its share of
    the workgroups, with its own pitch and LDS geometry), adding into `tables`.  items: [(batch, s_band, s_min)] as
    accumulate() takes them one by one -- same counts; bands the merged kernel does not serve are launched alone."""
    if not items:
        return
    dev_ = items[0][0].seq.device
    ctx = context(dev_.index)
    for batch, _, _ in items:
        if isinstance(batch, PairBatch) and tables.S2 != 2 * batch.S:
            raise ValueError('mate-pair rows of %d-base reads need tables with 2S = %d columns' % (batch.S, 2 * batch.S))
    dm = minscore if dinuc_minscore is None else dinuc_minscore
    arr = _band_array(items)
    N.check(N.load().kbbq_accumulate_bands_dev(ctx.handle, arr, len(items), tables.R, tables.S2, minscore, dm, N.ptr(tables.buf)))
    if check:
        ctx.status()


def apply_bands(items, lut_dev, shape, outs=None, minscore=MINSCORE, check=True, restore_order=False):
    """K2 over all length bands in one launch (kbbq_apply_bands_dev); returns the list of output planes (outs, or fresh
    ones).  The table-driven LUT only (a shape the fast kernels cannot serve raises LutNeedsCheckedApply before anything
    is launched: apply() band by band has the checked kernel)."""
    torch = _torch()
    R, Qt, S2, mode = shape
    if not items:
        return []
    if Qt != NQ or mode != N.APPLY_FAST:
        raise N.LutNeedsCheckedApply('the merged apply needs the fast LUT of a 43-row model')
    dev_ = items[0][0].seq.device
    ctx = context(dev_.index)
    lib = N.load()
    if outs is None:
        outs = [_take_out_plane(b) for b, _, _ in items]
    pluts = []
    for batch, _, _ in items:
        plut = None
        if isinstance(batch, PairBatch):
            if S2 != 2 * batch.S:
                raise N.LutNeedsCheckedApply('this layout needs the fast LUT of a %d-column model' % S2)
            key = (dev_.index, R, S2)
            plut = _pair_luts.get(key)
            if plut is None:
                plut = torch.empty(lib.kbbq_pair_lut_bytes(R, NQ, S2), dtype=torch.uint8, device=dev_)
                _pair_luts[key] = plut
            N.check(lib.kbbq_pair_lut_rows_dev(ctx.handle, N.ptr(lut_dev), R, S2, minscore, _row_flags(batch), N.ptr(plut)))
        pluts.append(plut)
    arr = _band_array(items, outs, pluts, restore_order)
    N.check(lib.kbbq_apply_bands_dev(ctx.handle, arr, len(items), R, S2, minscore, N.ptr(lut_dev)))
    if check:
        ctx.status()
    return outs


def build_lut(meanq, rgdq, qdq, posdq, dinucdq, minscore=MINSCORE):
    """Fold the five model arrays into the apply LUT blob (host, tiny).  Returns
    (blob uint8 ndarray, (R, Qt, S2, mode))."""
    a = [np.ascontiguousarray(np.asarray(x), dtype=np.int64) for x in (meanq, rgdq, qdq, posdq, dinucdq)]
    if a[3].ndim != 3 or a[4].ndim != 3:
        raise ValueError('positiondeltaq / dinucdeltaq must be 3-d')
    R, Qt, S2 = a[3].shape
    D = a[4].shape[2]
    if a[0].shape != (R,) or a[1].shape != (R,) or a[2].shape != (R, Qt) or a[4].shape[:2] != (R, Qt):
        raise IndexError('delta-Q tables have inconsistent shapes')
    lib = N.load()
    blob = np.zeros((lib.kbbq_lut_bytes(R, Qt, S2) + 7) // 8, dtype=np.int64).view(np.uint8)
    flags = ctypes.c_int(0)
    N.check(lib.kbbq_build_lut(R, Qt, S2, D, minscore, N.ptr(a[0]), N.ptr(a[1]), N.ptr(a[2]), N.ptr(a[3]),
                               N.ptr(a[4]), N.ptr(blob), ctypes.byref(flags)))
    return blob, (R, Qt, S2, N.APPLY_FAST if flags.value == 0 else N.APPLY_CHECKED)


def apply(batch, lut_dev, shape, out=None, minscore=MINSCORE, check=True, restore_order=False):
    """K2 over a device batch: new quality bytes [n, pitch] (compare_reads.py:320-328).
    With check=False the caller must call context().status() itself (and re-run with
    mode APPLY_CHECKED on LutNeedsCheckedApply).  restore_order: rows of a batch grouped by read group are stored
    through its `perm`, i.e. straight back into the row order before the grouping."""
    torch = _torch()
    R, Qt, S2, mode = shape
    ctx = context(batch.seq.device.index)
    if out is None:
        out = _take_out_plane(batch)
    pairs = isinstance(batch, PairBatch)
    grouped = batch.seg is not None
    if pairs or grouped or batch.nib:
        # these layouts use the table-driven LUT only (pair rows: their own layout of it, derived on the device from
        # the blob); rows the fast kernel cannot serve (and a LUT that is not range-safe) surface as
        # LutNeedsCheckedApply: re-run on plain one-read-per-row planes
        if Qt != NQ or mode != N.APPLY_FAST or (pairs and S2 != 2 * batch.S):
            raise N.LutNeedsCheckedApply('this layout needs the fast LUT of a %d-column model' % S2)
        lib = N.load()
        plut = None
        if pairs:
            key = (batch.seq.device.index, R, S2)
            plut = _pair_luts.get(key)
            if plut is None:
                plut = torch.empty(lib.kbbq_pair_lut_bytes(R, NQ, S2), dtype=torch.uint8, device=batch.seq.device)
                _pair_luts[key] = plut
            N.check(lib.kbbq_pair_lut_rows_dev(ctx.handle, N.ptr(lut_dev), R, S2, minscore, _row_flags(batch), N.ptr(plut)))
        perm = batch.perm if (restore_order and grouped) else None
        N.check(lib.kbbq_apply_rows_dev(ctx.handle, N.ptr(batch.seq), N.ptr(batch.qual), N.ptr(batch.meta), batch.n,
                                        batch.pitch, _row_flags(batch), R, S2, minscore, N.ptr(lut_dev), N.ptr(plut),
                                        N.ptr(batch.seg), N.ptr(perm), N.ptr(out)))
        if check:
            ctx.status()
        return out

    def launch(m):
        N.check(N.load().kbbq_apply_dev(ctx.handle, N.ptr(batch.seq), N.ptr(batch.qual), N.ptr(batch.meta),
                                        batch.n, batch.pitch, R, Qt, S2, minscore, N.ptr(lut_dev), m,
                                        N.ptr(out)))
    launch(mode)
    if check:
        try:
            ctx.status()
        except N.LutNeedsCheckedApply:
            launch(N.APPLY_CHECKED)
            ctx.status()
    return out


def lut_to_device(lut, device=None):
    torch = _torch()
    return torch.from_numpy(lut).to('cuda' if device is None else device)


_consts = None


def _model_consts():
    global _consts
    if _consts is None:
        from . import _solve
        _consts = _solve.model_consts()
    return _consts


def delta_q(prior_q, numerrs, numtotal):
    """K3, generic form: compare_reads.gatk_delta_q on the device (host: gammaln terms).
    An integer prior returns integers; a float prior (the report builder's EstimatedQReported,
    reference gatk/bqsr.py:294) returns `posterior - prior` in float64, as the reference does."""
    from . import _solve
    torch = _torch()
    prior = np.asarray(prior_q)
    is_float = prior.dtype.kind == 'f'
    pq = np.ascontiguousarray(prior, dtype=np.float64 if is_float else np.int64)
    e = np.ascontiguousarray(np.asarray(numerrs), dtype=np.int64)
    t = np.ascontiguousarray(np.asarray(numtotal), dtype=np.int64)
    assert pq.shape == e.shape == t.shape
    # prior_dist[|q' - prior_q|] has 43 entries: the distance to q' = 0 or 42 must stay below 43
    if pq.size and (not np.all(np.isfinite(pq)) or np.trunc(pq.max()) > MAXSCORE or np.trunc(pq.min() - MAXSCORE) < -MAXSCORE):
        worst = int(max(abs(np.trunc(0 - pq.max())), abs(np.trunc(MAXSCORE - pq.min())))) if np.all(np.isfinite(pq)) else -1
        raise IndexError('index %d is out of bounds for axis 0 with size %d' % (worst, MAXSCORE + 1))
    if pq.size == 0:
        return np.zeros(pq.shape, dtype=np.float64 if is_float else np.int_)
    comb = np.ascontiguousarray(_solve.combiln(e, t), dtype=np.float64)
    ctx = context()
    d = [torch.from_numpy(x.ravel()).cuda() for x in (pq, e, t, comb)]
    out = torch.empty(pq.size, dtype=torch.int64, device='cuda')
    fn = N.load().kbbq_posterior_q_dev if is_float else N.load().kbbq_delta_q_dev
    N.check(fn(ctx.handle, N.ptr(d[0]), N.ptr(d[1]), N.ptr(d[2]), N.ptr(d[3]),
               pq.size, N.ptr(_model_consts()), N.ptr(out)))
    res = out.cpu().numpy().reshape(pq.shape)
    return res - pq if is_float else res.astype(np.int_)


_logtabs = {}            # device index -> the host libm's log() constants on the device, or None (host pass in use)
_logtab_jobs = {}        # device index -> the thread that is establishing _logtabs[device] (warm_up)


def _prove_logtab(device):
    try:
        torch = _torch()
        with torch.cuda.device(device):
            _logtab_check(device)
    except Exception:                        # noqa: BLE001 -- no proof: the solve keeps its host pass
        _logtabs[device] = None
_solve_bufs = {}


def _gammaln_check_arguments():
    """> 10^6 integer-valued arguments of the kind the solve forms (counts + 1 ... + 3): every small one, a log-uniform
    sample up to 2^40, and the neighbourhoods of the routine's branch points and of the powers of two."""
    rng = np.random.default_rng(20240229)
    edges = np.array([1, 2, 3, 12, 13, 14, 999, 1000, 1001, 1e8 - 1, 1e8, 1e8 + 1, 1e10, 7.5e9 + 3], dtype=np.float64)
    pow2 = 2.0 ** np.arange(0, 41)
    return np.concatenate([np.arange(1, 262145, dtype=np.float64), np.floor(2.0 ** rng.uniform(0, 40, 1 << 20)),
                           edges, pow2, pow2[1:] + 1, pow2[2:] - 1])


def device_logtab(device=None):
    """The constants of the HOST libm's log() on the device (csrc/lgam_core.h), once the device's gammaln over them has
    been checked bit for bit against the host routine on > 10^6 arguments; None when the constants are not found or any
    bit differs (the solve then keeps its host pass).  KBBQ_HOST_SOLVE=1 forces None."""
    import os
    torch = _torch()
    if device is None:
        device = torch.cuda.current_device()
    job = _logtab_jobs.pop(device, None)
    if job is not None:                      # started by warm_up(): wait for its verdict
        job.join()
    if device in _logtabs:
        return _logtabs[device]
    return _logtab_check(device)


def _logtab_check(device):
    """The check itself (device_logtab's docstring); records and returns its verdict."""
    import os
    torch = _torch()
    tab = None
    if not os.environ.get('KBBQ_HOST_SOLVE'):
        lib = N.load()
        host = np.zeros(263, dtype=np.float64)
        if lib.kbbq_libm_log_data(N.ptr(host), host.size) == N.KBBQ_OK:
            with torch.cuda.device(device):
                ctx = context(device)
                cand = torch.from_numpy(host).cuda()
                x = _gammaln_check_arguments()
                want = np.empty_like(x)
                N.check(lib.kbbq_gammaln_host(N.ptr(x), x.size, N.ptr(want)))
                d_x = torch.from_numpy(x).cuda()
                d_out = torch.empty_like(d_x)
                N.check(lib.kbbq_gammaln_dev(ctx.handle, N.ptr(d_x), x.size, N.ptr(cand), N.ptr(d_out)))
                got = d_out.cpu().numpy()
                if np.array_equal(got.view(np.uint64), want.view(np.uint64)):
                    tab = cand
    _logtabs[device] = tab
    return tab


def _consts172():
    global _consts172_
    if _consts172_ is None:
        from . import compare_reads as utils
        perr = np.asarray(utils.q_to_p(np.arange(NQ, dtype=np.int_)), dtype=np.float64)
        _consts172_ = np.ascontiguousarray(np.concatenate([_model_consts(), perr]))
    return _consts172_


_consts172_ = None


def solve_lut(tables, minscore=MINSCORE, check=True, reuse=False):
    """(lut_dev, shape) for apply(): count tables -> apply LUT with NOTHING from the host inside the step when the
    device's gammaln has been proven equal to the host's (device_logtab): marginals, gammaln terms, meanq and the four
    levels of applybqsr.get_delta_qs are kernels on the launch stream, the buffers are kept per table shape.  A meanq the
    kernel cannot decide in double precision (within 1e-7 of a truncation boundary -- one quality value only is such a
    case) is reported by the status word: with check=True it is read here (one synchronisation) and the solve redone
    with the host's longdouble meanq; with check=False the caller's context().status() raises MeanqNeedsHost.
    reuse: the LUT is the buffer kept per (device, R, S2) -- the next solve_lut of that shape overwrites it in place (a step
    loop that solves and applies again and again wants exactly that); otherwise the caller gets a copy of its own."""
    torch = _torch()
    dev_ = tables.buf.device
    tab = device_logtab(dev_.index)
    if tab is None:
        return _host_fed_lut(tables, minscore, check)
    R, S2 = tables.R, tables.S2
    lib = N.load()
    key = (dev_.index, R, S2)
    bufs = _solve_bufs.get(key)
    if bufs is None:
        bufs = (torch.zeros(lib.kbbq_lut_bytes(R, NQ, S2), dtype=torch.uint8, device=dev_),
                torch.empty(R * NQ, dtype=torch.int32, device=dev_))
        _solve_bufs[key] = bufs
    lut, post_q = bufs
    ctx = context(dev_.index)
    N.check(lib.kbbq_solve_device_dev(ctx.handle, N.ptr(tables.buf), R, S2, minscore, N.ptr(tab), N.ptr(_consts172()),
                                      N.ptr(post_q), N.ptr(lut), None, None))
    if check:
        try:
            ctx.status()
        except N.MeanqNeedsHost:
            return _host_fed_lut(tables, minscore, True)
        except N.LutNeedsCheckedApply:
            # k3_fill_full_lut found a value that does not fit int8 or a (cycle, context) sum that can leave 0..255: the
            # canonical int16 part of the blob is complete, the table-driven part is not usable -- the checked kernel
            # serves this LUT (apply() refuses the layouts that need the table-driven part and the caller falls back to
            # one read per row, as with a host-built blob whose flags are set)
            return (lut if reuse else lut.clone()), (R, NQ, S2, N.APPLY_CHECKED)
    return (lut if reuse else lut.clone()), (R, NQ, S2, N.APPLY_FAST)


def _host_fed_lut(tables, minscore, check):
    """solve() for solve_lut's callers: (lut, shape), the shape's mode taken from the status word the LUT builder
    (k3_fill_full_lut) sets when the table-driven part cannot serve the LUT -- read here when `check`, else left for the
    caller's context().status() as in solve_lut."""
    lut, shape, _, _ = solve(tables, minscore=minscore)
    if check:
        try:
            context(tables.buf.device.index).status()
        except N.LutNeedsCheckedApply:
            shape = shape[:3] + (N.APPLY_CHECKED,)
    return lut, shape


def solve(tables, want_dq=False, minscore=MINSCORE):
    """K3, fused form: count tables (device) -> apply LUT (device), the whole of
    applybqsr.get_delta_qs.  Host work: marginals, meanq and the gammaln terms.
    Returns (lut_dev, shape, vectors, dqs) -- dqs is None unless want_dq."""
    from . import _solve
    torch = _torch()
    R, S2 = tables.R, tables.S2
    # host half in one native pass over the buffer as it comes off the device: marginals (recalibrate.py:112-115) and the
    # gammaln term of every cell; meanq (longdouble, 43 terms per read group) stays NumPy
    h = tables.buf.cpu().numpy()
    aux, q_e, q_t, rg_e, rg_t = _solve.solve_prep(h, R, S2)
    npos, ndn = R * NQ * S2, R * NQ * 16
    p_e, p_t = h[:npos].reshape(R, NQ, S2), h[npos:2 * npos].reshape(R, NQ, S2)
    d_e, d_t = h[2 * npos:2 * npos + ndn].reshape(R, NQ, 16), h[2 * npos + ndn:].reshape(R, NQ, 16)
    meanq = _solve.meanq_from_q_total(q_t)
    vectors = (meanq, rg_e, rg_t, q_e, q_t, p_e, p_t, d_e, d_t)
    lib = N.load()
    assert aux.size == lib.kbbq_solve_aux_count(R, S2)
    dev = tables.buf.device
    # one upload: [aux float64 | meanq int32]
    host = np.concatenate([np.ascontiguousarray(aux, dtype=np.float64).view(np.uint8),
                           np.ascontiguousarray(meanq, dtype=np.int32).view(np.uint8)])
    d_host = torch.from_numpy(host).to(dev)
    d_aux, d_meanq = d_host[:aux.size * 8], d_host[aux.size * 8:]
    post_q = torch.empty(R * NQ, dtype=torch.int32, device=dev)
    lut = torch.zeros(lib.kbbq_lut_bytes(R, NQ, S2), dtype=torch.uint8, device=dev)
    dq = torch.empty(lib.kbbq_solve_dq_count(R, S2), dtype=torch.int32, device=dev) if want_dq else None
    ctx = context(dev.index)
    N.check(lib.kbbq_solve_dev(ctx.handle, N.ptr(tables.buf), R, S2, minscore, N.ptr(d_meanq), N.ptr(d_aux),
                               N.ptr(_model_consts()), N.ptr(post_q), N.ptr(lut), N.ptr(dq)))
    dqs = None
    if want_dq:
        h = dq.cpu().numpy().astype(np.int_)
        o1, o2, o3 = R, R + R * NQ, R + R * NQ + R * NQ * S2
        dqs = (h[:o1].copy(), h[o1:o2].reshape(R, NQ).copy(), h[o2:o3].reshape(R, NQ, S2).copy(),
               h[o3:].reshape(R, NQ, 17).copy())
    return lut, (R, NQ, S2, N.APPLY_FAST), vectors, dqs
