"""
The file path in CONSTANT device memory (and, for inputs read sequentially, constant host memory).

The reference walks its two files read by read (recalibrate.py:56-57 pass 1, :141-156 pass 2) and so takes inputs of any
size.  The resident file path (recalibrate._pack_and_tally) keeps a rank's whole shard on the device between pass 1 and
pass 2 -- fastest while it fits (a second fill of file A is saved), a wall at about 600 M x 150 bp reads per GPU.  Here a
shard larger than the device budget (_device.device_budget: KBBQ_DEVICE_BUDGET, else 60 % of the free memory) goes through
the same kernels SLAB BY SLAB:

    pass 1   fill (host threads, the packer writes the device layout) | H2D | K1 into the same count tables (K1 adds)
    pass 2   fill file A again | H2D | K2 | D2H | format | write        (_egress.emit_produced: the producer is a generator)

A slab is a run of reads of one length band, laid out exactly as a resident band would be (_device.laid_from_reader:
mate-pair rows, 4-bit planes, rows gathered by read-group segment -- decided per slab from its own sidecar statistics), so
every layout, fallback and error rule of the resident path applies slab-wise: whatever a slab's fast kernels refuse is
redone on one character row per read.  Device memory: one slab of input in pass 1; a slab of input plus at most three
output planes in pass 2.

tally_range / produce_range work on any reader pair: the mapped, indexed readers of a regular file (a rank's byte range
included) and the segments of a sequentially read input (fastx.FastqStream: pipes, process substitutions, stdin).
"""
from . import _device as dev
from . import fastx
from ._trace import stage

REFUSALS = (IndexError, TypeError, ValueError, dev.N.LutNeedsCheckedApply)


def resident_bytes(n, S):
    """Upper bound of what the resident path holds on the device for n reads of up to S bases: three input planes, the
    output plane, the sidecars."""
    return int(n) * (4 * fastx.pitch_for(S) + 4)


def slab_reads(budget, bytes_per_read):
    """Reads per slab (even: a slab of mate-pair rows starts at a first mate; at least 2)."""
    return max(2, int(budget * 0.8) // max(int(bytes_per_read), 1) & ~1)


class Peak:
    """What a streamed run held on the device at most (bytes of tensors), for recalibrate.LAST_RUN."""

    def __init__(self, budget):
        self.budget = int(budget)
        torch = dev._torch()
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        dev.memory_peak(reset=True)
        self.before = int(torch.cuda.memory_allocated()) if hasattr(torch.cuda, 'memory_allocated') else 0
        if hasattr(torch.cuda, 'set_reserve_limit'):          # kbbq/_hipmem.py: released slabs are kept for the next one
            torch.cuda.set_reserve_limit(self.before + self.budget)

    def report(self):
        return {'device_budget_bytes': self.budget, 'resident_before_bytes': self.before,
                'peak_device_bytes': max(dev.memory_peak() - self.before, 0)}

    def close(self):
        torch = dev._torch()
        if hasattr(torch.cuda, 'set_reserve_limit'):
            torch.cuda.set_reserve_limit(None)


def _slab_band(A, B, infer_rg, origin, s_lo, m, longest, shortest, pitch):
    """The dictionary the resident path keeps per length band (fastx._fill_bands), for ONE slab: `first` counts from `origin`
    (the rank's or the segment's first read), `source` lets fastx.band_rows fill character rows on demand."""
    return dict(first=s_lo, n=m, S=longest, Smin=shortest, pitch=pitch, source=(A, B, infer_rg, origin + s_lo), batch=None, laid=None,
                keep_pinned=True)


def _tally_slabs(A, B, infer_rg, lo, hi, tables, minscore, budget, careful):
    R, S2 = tables.R, tables.S2
    for b_lo, b_hi, longest, shortest in A.length_bands(lo, hi - lo):
        pitch = fastx.pitch_for(longest)
        step = slab_reads(budget, 3 * pitch + 4)                 # character rows, three planes: the widest a slab gets
        hints = dict(s_band=longest, s_min=shortest)
        for s_lo in range(b_lo, b_hi, step):
            m = min(step, b_hi - s_lo)
            band = _slab_band(A, B, infer_rg, lo, s_lo, m, longest, shortest, pitch)
            with stage('fill'):
                laid = dev.laid_from_reader(A, B, infer_rg, lo + s_lo, m, pitch, max(R, 1), packed=longest <= dev.PACKED_READS,
                                            pair_S=S2 // 2, keep_pinned=True)
            try:
                done = False
                if laid is not None:
                    try:
                        with stage('K1'):
                            if careful:                          # a refused attempt must leave nothing behind in the tables
                                part = dev.Tables(R, S2)
                                dev.accumulate(laid, part, minscore, **hints)
                                tables.add(part)
                            else:
                                dev.accumulate(laid, tables, minscore, check=False, **hints)
                        done = True
                    except REFUSALS:
                        pass                                     # bad input or an unsupported shape: the row-per-read kernel decides
                    laid = None
                if not done:
                    with stage('fill'):
                        rows = fastx.band_rows(band)
                    with stage('K1'):
                        dev.accumulate(rows, tables, minscore, check=careful, **hints)
                    rows = None
            except (IndexError, TypeError) as e:
                if hasattr(e, 'read_index'):
                    e.read_index = s_lo + max(e.read_index, 0)
                raise
            finally:
                band['batch'] = None


def tally_range(A, B, infer_rg, lo, hi, tables, minscore, budget):
    """K1 over reads [lo, hi) of the reader pair (A, B) -- record numbers as the readers count them -- slab by slab, adding
    into `tables`.  First without a look at the kernels' status until the end (no wait per slab); when anything was flagged
    the tables are put back and the range is walked again slab by slab with every fallback and the first offending read
    found, its index counted from `lo` (as recalibrate._tally_local reports it)."""
    if hi <= lo:
        return
    ctx = dev.context(tables.buf.device.index)
    before = tables.buf.clone()
    try:
        try:
            _tally_slabs(A, B, infer_rg, lo, hi, tables, minscore, budget, careful=False)
            with stage('K1', sync=True):
                ctx.status()
            return
        except REFUSALS:
            try:
                ctx.status()                                     # whatever else is pending belongs to the abandoned attempt
            except Exception:                                    # noqa: BLE001
                pass
            tables.buf.copy_(before)
        _tally_slabs(A, B, infer_rg, lo, hi, tables, minscore, budget, careful=True)
    finally:
        dev.release_pinned('ingest')


def apply_band(band, lut, shape):
    """New qualities of a band (or of a slab of one): in the band's own layout (mate-pair rows stay mate-pair rows, stored
    in input order -- the writer reads them as they are) or, when the layout's kernel cannot serve the LUT or the rows, one
    read per row from the checked kernel.  Sets band['out_flags'] / ['out_S2'] for the writer."""
    laid, out = band.get('laid'), None
    band['out_flags'], band['out_S2'] = 0, 0
    if laid is not None:
        try:
            out = dev.apply(laid, lut, shape, restore_order=True)      # grouped rows: stored straight back in input order
            if isinstance(laid, dev.PairBatch):
                band['out_flags'], band['out_S2'] = dev.N.ROWS_PAIRS, 2 * laid.S
        except dev.N.LutNeedsCheckedApply:
            out = None                   # a LUT the fast kernel cannot serve: the checked row-per-read kernel
    if out is None:
        band['out_flags'], band['out_S2'] = 0, 0
        out = dev.apply(fastx.band_rows(band), lut, shape)
    return out


def produce_range(A, infer_rg, lo, hi, lut, shape, budget, origin=None, extra=None, device=None):
    """Generator of (band, device plane of new qualities) over reads [lo, hi) of reader A, slab by slab: fill | H2D | K2, each
    slab's input dropped as soon as its kernel has run (the plane lives until the egress pipeline has copied it).  Meant to
    be consumed by _egress.emit_produced on its feeding thread.  band['first'] counts from `origin` (default lo); extra: keys
    every band gets (a segment's 'text' and 'base').  A kernel-reported error carries the read's index counted from origin."""
    torch = dev._torch()
    R, Qt, S2, mode = shape
    origin = lo if origin is None else origin
    with torch.cuda.device(torch.cuda.current_device() if device is None else device):
        try:
            for b_lo, b_hi, longest, shortest in A.length_bands(lo, hi - lo):
                pitch = fastx.pitch_for(longest)
                step = slab_reads(budget, 5 * pitch + 4)             # a slab's input (<= 2 planes) + its plane and two more in the pipeline
                for s_lo in range(b_lo, b_hi, step):
                    m = min(step, b_hi - s_lo)
                    band = _slab_band(A, None, infer_rg, lo, s_lo, m, longest, shortest, pitch)
                    band['first'] = lo + s_lo - origin
                    if extra:
                        band.update(extra)
                    with stage('fill'):
                        try:
                            band['laid'] = dev.laid_from_reader(A, None, infer_rg, lo + s_lo, m, pitch, max(R, 1),
                                                                packed=longest <= dev.PACKED_READS, pair_S=S2 // 2, keep_pinned=True)
                        except ValueError:                           # a read group the model does not have (recalibrate.py:143-151):
                            band['laid'] = None                      # one read per row decides -- an IndexError at the first such read
                    try:
                        with stage('apply'):
                            out = apply_band(band, lut, shape)
                    except REFUSALS as e:
                        if hasattr(e, 'read_index'):
                            e.read_index = band['first'] + max(e.read_index, 0)
                        raise
                    band['laid'] = band['batch'] = None               # the input planes go back to the allocator
                    yield band, out
                    del out
        finally:
            dev.release_pinned('ingest')
