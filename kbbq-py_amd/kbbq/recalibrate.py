"""
kbbq.recalibrate -- drop-in for the reference module of the same name
(reference kbbq/recalibrate.py:13-174), with both per-read Python loops replaced
by HIP kernels on an MI355X:

    pass 1  fastq_to_covariate_arrays  -> K1 (error flagging + covariate binning)
    model   applybqsr.get_delta_qs     -> host, same SciPy call as the reference
    pass 2  recalibrate_fastq          -> K2 (delta-Q LUT apply), FASTQ text to stdout

Behaviour kept: argument names and defaults, the 9-tuple order, int64 arrays
owned by the caller, output through print(), and the exceptions the reference
raises on bad input (AssertionError, IndexError, TypeError, NotImplementedError,
ValueError -- SURVEY.md 8(b)).

Deliberate deviations (also in README.md / INTEGRATION.md):
  * gatkreport: the reference raises NotImplementedError for ANY value (recalibrate.py:167-168); here `-g FILE` with FASTQ
    input saves / loads the model; without FASTQ input it raises NotImplementedError as the reference.
  * maxscore: the device tables have the reference's default Q axis (43 = maxscore 42 + 1); another maxscore raises
    ValueError instead of being honoured.
  * output=...: an extra keyword (the reference only prints): write to a file, one file per rank under torch.distributed.
"""
import os

import numpy as np

from . import compare_reads as utils        # noqa: F401  (module attribute of the reference: recalibrate.utils)
from . import fastx
from . import _device as dev
from . import _solve
from . import _stream
from . import parallel
from ._trace import stage
from .gatk import applybqsr


def find_corrected_sites(uncorr_read, corr_read):
    """Boolean array, True where the corrected sequence differs from the original."""
    assert corr_read.name.startswith(uncorr_read.name)
    a = np.frombuffer(uncorr_read.sequence.encode('utf-32-le'), dtype=np.uint32)
    b = np.frombuffer(corr_read.sequence.encode('utf-32-le'), dtype=np.uint32)
    if a.shape != b.shape:
        raise ValueError('operands could not be broadcast together with shapes %s %s' % (a.shape, b.shape))
    return a != b


def _vectors_from_tables(pos_errs, pos_total, dinuc_errs, dinuc_total, maxscore):
    return _solve.vectors_from_tables(pos_errs, pos_total, dinuc_errs, dinuc_total, maxscore)


LAST_RUN = {}            # what the most recent recalibrate_fastq did per length band (layouts, bytes uploaded): bench.py, traces


def _tally_local(packed, minscore, maxscore):
    """K1 over this rank's packed reads, length band by length band -> device Tables (zero tables when the rank has
    no reads).  Each band is tallied in the layout the packer wrote it in (fastx._fill_bands); whatever that path reports
    (bad input, a shape it does not serve) is redone one read per row, which carries the reference's exact error semantics.  The read index
    of a kernel-reported error is made relative to the rank's first read."""
    if maxscore != 42:
        raise ValueError('the Q axis of the device tables is fixed at 43 (maxscore = 42)')
    R, S = max(packed['R'], 0), packed['S']
    if R == 0 or S == 0:
        return None
    if packed.get('streamed'):
        # a shard larger than the device budget: nothing is resident, the reads go through K1 slab by slab (kbbq/_stream.py)
        st = packed['streamed']
        st['peak'] = _stream.Peak(st['budget'])
        tables = dev.Tables(R, 2 * S)
        _stream.tally_range(st['A'], st['B'], st['infer_rg'], st['lo'], st['hi'], tables, minscore, st['budget'])
        return tables
    tables = dev.Tables(R, 2 * S)

    def tally_band(band, laid):
        hints = dict(s_band=band['S'], s_min=band.get('Smin', 0))
        if laid is not None:
            part = dev.Tables(R, 2 * S)
            try:
                dev.accumulate(laid, part, minscore, **hints)
                tables.add(part)
                band['laid'] = laid
                return
            except (IndexError, TypeError, ValueError, dev.N.LutNeedsCheckedApply):
                band['laid'] = None          # bad input or an unsupported shape: the row-per-read kernel decides
        dev.accumulate(fastx.band_rows(band), tables, minscore, **hints)

    # several length bands (a mixed-length input, BASELINE config 5): ONE launch over all of them, each band on its share of
    # the workgroups (kbbq_accumulate_bands_dev); whatever any band's kernel reports sends them all band by band below, where
    # each has its own fallback and the first offending read is found
    done = set()
    several = [b for b in packed['bands'] if 'source' in b and b.get('laid') is not None]
    if len(several) >= 2:
        part = dev.Tables(R, 2 * S)
        try:
            with stage('K1', sync=True):
                dev.accumulate_bands([(b['laid'], b['S'], b.get('Smin', 0)) for b in several], part, minscore)
            tables.add(part)
            done = {id(b) for b in several}
        except (IndexError, TypeError, ValueError, dev.N.LutNeedsCheckedApply):
            pass
    for band in packed['bands']:
        if id(band) in done:
            continue
        if 'source' in band:
            laid = band.get('laid')                    # written by the packer in its layout (fastx._fill_bands): nothing to convert
        else:                                          # host planes of a caller's own: tallied and applied as they are -- a
            with stage('H2D', sync=True):              # device pass into another layout costs more than it saves for ONE
                band['batch'] = dev.ReadBatch.from_host(band['seq'], band['qual'], band['meta'], cseq=band['cseq'])   # accumulate + apply
            laid = None
        band['laid'] = None                            # still resident: pass 2 re-uses them when it covers file A
        try:
            with stage('K1', sync=True):
                tally_band(band, laid)
        except (IndexError, TypeError) as e:
            if hasattr(e, 'read_index'):
                e.read_index = band['first'] + max(e.read_index, 0)
            raise
    return tables


def _tally(packed, minscore, maxscore):
    """_tally_local on every rank, agreement on the first error any rank's kernel reported (all ranks
    raise it), then ONE sum-allreduce of the count tables (RCCL over xGMI)."""
    exc, tables = None, None
    try:
        tables = _tally_local(packed, minscore, maxscore)
    except Exception as e:                   # noqa: BLE001 -- whatever stops one rank (bad input, a device error, out of
        exc = e                              # memory) must stop them all before anybody waits in the allreduce
    parallel.raise_first_error(exc, None if exc is None else packed.get('first', 0) + max(getattr(exc, 'read_index', 0), 0))
    if tables is not None:
        parallel.allreduce_tables(tables.buf)
    return tables


def _warm_up():
    """The process's one-time device costs (dev.warm_up), paid while the C++ reader works on the input files."""
    import sys
    torch = dev._backend or sys.modules.get('torch')         # the device this process selected (parallel.init_from_env), else 0
    device = torch.cuda.current_device() if torch is not None and torch.cuda.is_available() else 0
    with stage('warm-up'):
        dev.warm_up(device)


def _pack_and_tally(fastq, infer_rg, minscore, maxscore):
    world, rank = parallel.world_rank()
    # rank 0 (or the only process) opens and scans the pair on the reader's own threads -- no interpreter lock needed --
    # while the device warms up; the other ranks then index only the byte ranges of their shards (fastx.pack_pair)
    # (several ranks: every rank cuts, indexes and scans its own byte range of either file -- fastx.local_ranges -- and
    # only when the files do not cut alike does rank 0 scan all of them for a plan)
    scan = fastx.PairScan(fastq[0], fastq[1], infer_rg) if world == 1 else None
    _warm_up()
    failure, packed = None, None
    try:
        packed = fastx.pack_pair(fastq[0], fastq[1], infer_rg, shard=(rank, world) if world > 1 else None, bands=True,
                                 scan=scan, to_device=True, exchange=parallel.broadcast_object if world > 1 else None,
                                 gather=parallel.all_gather_object if world > 1 else None, budget=dev.device_budget())
    except Exception as e:                   # noqa: BLE001 -- one rank's failure (its shard unreadable, out of memory) stops them all
        failure = e
    parallel.raise_first_error(failure, 0)
    err = packed.get('pending_error')
    if err is not None:
        # the reference fails at the FIRST offending read: let the kernel look at the reads
        # before it (and at it, when its own checks come first) before raising the host error
        idx, exc, inclusive = err
        try:
            _tally(packed, minscore, maxscore)
        finally:
            if packed.get('other') is not None:
                fastx.close_later(packed.pop('other'))
        raise exc
    try:
        tables = _tally(packed, minscore, maxscore)
    finally:
        # the corrected file is not needed past the tally.  It is only RETIRED here (no new work starts on it) and unmapped by the
        # caller after its last byte: unmapping 2.5 GB holds the address space's lock for 30-40 ms, and done beside pass 2 --
        # as rounds 3 and 4 did -- it stalled the first slabs' page faults, buffer allocations and write(2) for as long
        if packed.get('other') is not None:
            packed['retired'] = fastx.retire(packed.pop('other'))
    if tables is not None and packed.get('total', packed['n']) == 0:
        tables = None
    return packed, tables


def fastq_to_covariate_arrays(fastq, infer_rg=False, minscore=6, maxscore=42):
    """Tally errors and observations of the (uncorrected, corrected) FASTQ pair by read
    group, reported quality, cycle and dinucleotide.  Returns the reference's 9-tuple:
    meanq, rg_errs, rg_total, q_errs, q_total, pos_errs, pos_total, dinuc_errs, dinuc_total."""
    if any(fastx.is_sequential_input(p) for p in fastq) or os.environ.get('KBBQ_SEQUENTIAL'):
        if maxscore != 42:
            raise ValueError('the Q axis of the device tables is fixed at 43 (maxscore = 42)')
        run = _Sequential(fastq, infer_rg)
        try:
            _warm_up()
            tables = run.pass1(minscore, spool=False)
        finally:
            run.close()
        packed = dict(R=len(run.rgs) if run.usable else 0, S=run.longest)
    else:
        packed, tables = _pack_and_tally(fastq, infer_rg, minscore, maxscore)
        if packed.get('retired') is not None:
            fastx.close_later(packed.pop('retired'))
    if tables is None:
        z = lambda *s: np.zeros(s, dtype=np.int64)
        R, S = packed['R'], packed['S']
        return _vectors_from_tables(z(R, 43, 2 * S), z(R, 43, 2 * S), z(R, 43, 16), z(R, 43, 16), maxscore)
    return _vectors_from_tables(*tables.to_host(), maxscore)


def _rg_order(rg_to_int):
    """Read-group names in id order, as text (the report's ReadGroup column)."""
    return [str(k) for k, _ in sorted(rg_to_int.items(), key=lambda kv: kv[1])]


def save_model(tables, rg_to_int, path):
    """Count tables (K1 output) -> GATK recalibration report file (gatk/bqsr.py
    vectors_to_report; the option main.py:55-58 declares)."""
    from .gatk import bqsr
    vectors = _vectors_from_tables(*tables.to_host(), 42)
    bqsr.vectors_to_report(*vectors, _rg_order(rg_to_int)).write(path)


def load_model(path, rg_to_int):
    """GATK recalibration report file -> device count tables for the given read groups
    (gatk/applybqsr.py table_to_vectors).  meanq is NOT taken from the 4-decimal
    EstimatedQReported column: K3 recomputes it from the counts exactly as pass 1 would have,
    so a saved-and-reloaded model recalibrates identically to a fresh one."""
    from . import recaltable
    report = recaltable.RecalibrationReport.fromfile(path)
    vectors = applybqsr.table_to_vectors(report, _rg_order(rg_to_int))
    return dev.Tables.from_host(*vectors[5:9])


def _collective(fn, first=0):
    """fn() on every rank; if a kernel of any rank flagged bad input, all ranks raise the error of the
    globally first offending read (a single process would have stopped there)."""
    exc, res = None, None
    try:
        res = fn()
    except Exception as e:                   # noqa: BLE001 -- see _tally
        exc = e
    parallel.raise_first_error(exc, None if exc is None else first + max(getattr(exc, 'read_index', 0), 0))
    return res


def recalibrate_fastq(fastq, infer_rg=False, gatkreport=None, output=None):
    """Recalibrate FASTQ file fastq[0] using its error-corrected version fastq[1];
    the recalibrated FASTQ is printed to stdout.  K1 -> K3 -> K2, tables and LUT stay on
    the device between the kernels.  gatkreport (the reference declares the option and raises
    NotImplementedError, recalibrate.py:167-168): an existing report replaces pass 1 (fastq[1]
    is not read); otherwise the model of pass 1 is saved there.
    Under torch.distributed (one process per GPU, parallel.init_from_env) every rank takes a contiguous
    shard of the records, the count tables are summed with one allreduce, the solve is replicated and
    the ranks print their records in rank order: the concatenated output is the single-process output.
    output (not in the reference, which only prints): write the records to this file instead of stdout; with several
    ranks every rank writes ITS records to `output`.rankNNNN at the same time (the files, concatenated in rank order,
    are the single-process output) -- ranks sharing one stdout can only write in turn."""
    world, rank = parallel.world_rank()
    if any(fastx.is_sequential_input(p) for p in fastq) or os.environ.get('KBBQ_SEQUENTIAL'):
        return _recalibrate_sequential(fastq, infer_rg, gatkreport, output)
    shard = (rank, world) if world > 1 else None
    done_with = []                           # readers to unmap on a helper thread once the last byte is out (or the call has failed)
    try:
        return _recalibrate_mapped(fastq, infer_rg, gatkreport, output, world, rank, shard, done_with)
    finally:
        for reader in done_with:
            if reader is not None:
                fastx.close_later(reader)


def _recalibrate_mapped(fastq, infer_rg, gatkreport, output, world, rank, shard, done_with):
    """recalibrate_fastq on mapped, indexed inputs (regular files): resident on the device, or slab by slab within the budget."""
    packed, single = None, None
    if gatkreport is not None and os.path.exists(gatkreport):
        scan = fastx.PairScan(fastq[0], None, infer_rg)
        _warm_up()
        text = scan.result()[0]
        done_with.append(text)
        if text.n == 0:
            return
        single = fastx.pack_single(text, infer_rg, shard, bands=True, to_device=True, budget=dev.device_budget())
        tables = load_model(gatkreport, single['rg_to_int'])
    else:
        packed, tables = _pack_and_tally(fastq, infer_rg, 6, 42)
        text = packed['text']
        done_with.extend([text, packed.get('retired')])
        if text.total == 0:
            return
        if tables is None:
            raise IndexError('index 0 is out of bounds for axis 0 with size 0')   # no read was tallied
        if gatkreport is not None:
            if rank == 0:
                save_model(tables, packed['rg_to_int'], gatkreport)
            parallel.barrier()
    with stage('solve', sync=True):
        lut, shape = dev.solve_lut(tables)
    R = shape[0]
    if packed is not None and packed['total'] == text.total:
        # pass 2 walks the same reads with the same first-appearance read groups (:141-148):
        # the planes of pass 1 are still on the device
        single = packed
    else:
        # file B was shorter (zip truncation), or the model came from a report: pass 2 covers
        # all of file A with its own first-appearance read groups
        if single is None:
            if text.n != text.total:
                text = fastx.NativeFastq(fastq[0])                 # a shard's reader: pass 2 needs all of file A
                done_with.append(text)
            single = fastx.pack_single(text, infer_rg, shard, bands=True, to_device=True, budget=dev.device_budget())

    if single.get('streamed'):
        return _emit_streamed(text, single, lut, shape, output, world, rank)

    if single['R'] != R:
        # pass 2 met read groups the model does not have (recalibrate.py:143-151: an IndexError at the first such read) or
        # fewer than it has: rows gathered by THEIR read-group segments do not fit the model's -- one read per row decides
        for band in single['bands']:
            band['laid'] = None

    def apply_band(band):
        return _stream.apply_band(band, lut, shape)

    def apply_shard():
        merged = {}
        several = [b for b in single['bands'] if b.get('laid') is not None]
        if len(several) >= 2:                 # all length bands in one launch (kbbq_apply_bands_dev); any trouble: band by band
            try:
                res = dev.apply_bands([(b['laid'], b['S'], b.get('Smin', 0)) for b in several], lut, shape, restore_order=True)
                for b, o in zip(several, res):
                    merged[id(b)] = o
                    two = isinstance(b['laid'], dev.PairBatch)
                    b['out_flags'], b['out_S2'] = (dev.N.ROWS_PAIRS, 2 * b['laid'].S) if two else (0, 0)
            except (IndexError, TypeError, ValueError, dev.N.LutNeedsCheckedApply):
                merged = {}
        outs = []
        for band in single['bands']:
            if id(band) in merged:
                outs.append(merged[id(band)])
                continue
            try:
                outs.append(apply_band(band))
            except (IndexError, TypeError, ValueError) as e:
                if hasattr(e, 'read_index'):
                    e.read_index = band['first'] + max(e.read_index, 0)
                raise
        return outs
    with stage('apply', sync=True):
        outs = _collective(apply_shard, single['first'])
    LAST_RUN.clear()
    LAST_RUN['bands'] = [dict(reads=b['n'], longest=b['S'],
                              layout=(b['laid'] if b.get('laid') is not None else b['batch']).describe(),
                              written_by='the FASTQ packer (kbbq_fastq_fill_rows)' if b.get('laid') is not None and 'source' in b else
                                         'the row-per-read packer (kbbq_fastq_fill_range)',
                              h2d_bytes=sum(getattr(x, 'h2d_bytes', 0) for x in (b.get('laid'), b.get('batch')) if x is not None),
                              output_rows='mate-pair rows, read by the writer as they are' if b.get('out_flags') else 'one read per row')
                         for b in single['bands']]

    # recalibrate.py:153-156: '@' + name, sequence, '+', qualities
    from . import _egress
    if output is None:
        parallel.in_rank_order(lambda: _egress.emit_records(text, single['first'], single['bands'], outs))
    else:
        with open(output if world == 1 else '%s.rank%04d' % (output, rank), 'wb') as sink:
            _egress.emit_records(text, single['first'], single['bands'], outs, sink=sink)


def _prefetched(items, depth=1):
    """Iterate `items` (a generator whose steps spend their time in C calls that release the interpreter lock: reading and
    indexing the next segment of an input) on a helper thread, `depth` items ahead of the consumer."""
    import queue
    import threading
    q, end = queue.Queue(depth), object()
    stop = threading.Event()

    def run():
        try:
            for it in items:
                if stop.is_set():
                    break
                q.put((it, None))
        except BaseException as e:           # noqa: BLE001 -- re-raised by the consumer
            q.put((None, e))
            return
        q.put((end, None))
    th = threading.Thread(target=run, daemon=True)
    th.start()
    try:
        while True:
            it, exc = q.get()
            if exc is not None:
                raise exc
            if it is end:
                return
            yield it
    finally:
        stop.set()
        while th.is_alive():                 # let a producer blocked on the full queue see the stop
            try:
                q.get_nowait()
            except queue.Empty:
                th.join(0.01)


SEGMENT_BYTES = 256 << 20                    # text of file A per segment of a sequentially read input (KBBQ_SEGMENT_BYTES)


def _segment_bytes():
    env = os.environ.get('KBBQ_SEGMENT_BYTES')
    return max(dev.parse_bytes(env), 1 << 16) if env else SEGMENT_BYTES


def _pair_segments(sa, sb, seg_bytes):
    """(segment of file A, segment of file B holding the same number of records -- fewer, or None, once file B has ended)
    until file A ends; without sb: (segment of A, None)."""
    import threading
    b_ended = sb is None
    while True:
        ahead = None
        if not b_ended:                      # both inputs are drained at the same time (two pipes fed by two decompressors)
            ahead = threading.Thread(target=_quietly, args=(sb.prefetch, seg_bytes), daemon=True)
            ahead.start()
        try:
            with stage('read'):
                a, _ = sa.next(seg_bytes)
        finally:
            if ahead is not None:
                ahead.join()
        if a is None:
            return
        b = None
        if not b_ended:
            with stage('read'):
                b, _ = sb.next(len_bytes(a) + (1 << 16), a.n)
            if b is None or b.n < a.n:
                b_ended = True
        yield a, b


def _quietly(fn, *args):
    """fn(*args) on a helper thread: a failure is left for the main call on the same stream to meet and report."""
    try:
        fn(*args)
    except Exception:                        # noqa: BLE001
        pass


def len_bytes(reader):
    """Bytes of text a segment holds (what its follower is asked for first)."""
    return max(reader.record_offset(reader.first + reader.n) - reader.record_offset(reader.first), 1 << 16) if reader.n else 1 << 16


def _grown(tables, R, S2):
    """`tables` at (at least) R read groups and S2 cycle columns: the reference appends zeros at the END of either axis when
    a longer read or a new read group arrives (recalibrate.py:66-87), so every count keeps its absolute index."""
    if tables is not None and tables.R >= R and tables.S2 >= S2:
        return tables
    if tables is None:
        return dev.Tables(R, S2)
    R, S2 = max(R, tables.R), max(S2, tables.S2)
    old = tables.to_host()
    new = [np.zeros((R, 43, S2), dtype=np.int64), np.zeros((R, 43, S2), dtype=np.int64),
           np.zeros((R, 43, 16), dtype=np.int64), np.zeros((R, 43, 16), dtype=np.int64)]
    for o, w in zip(old, new):
        w[:o.shape[0], :, :o.shape[2]] = o
    return dev.Tables.from_host(*new)


class _Sequential:
    """The two passes over inputs that are READ SEQUENTIALLY (fastx.FastqStream): pipes, process substitutions, standard
    input -- and regular files when KBBQ_SEQUENTIAL is set: no mapping, no whole-file index, host memory a few segments
    whatever the input's size.  What the reference's walk carries from read to read (recalibrate.py:59-101: read groups in
    first-appearance order, the longest read so far, the growing count arrays) is carried from segment to segment; the
    first offending read stops the walk where the reference stops.  File A is needed twice: a regular file is read again,
    anything else is copied to a spool file while pass 1 reads it (KBBQ_SPOOL_DIR, default the temporary directory)."""

    def __init__(self, fastq, infer_rg):
        world, _ = parallel.world_rank()
        if world > 1:
            raise ValueError('inputs that are read sequentially (pipes, standard input) need a single process: the ranks of a '
                             'multi-GPU run cut their byte ranges out of regular files')
        self.fastq, self.infer_rg = fastq, infer_rg
        self.spool = None
        self.rgs, self.longest, self.total_a, self.usable = [], 0, 0, 0

    def rg_to_int(self):
        return {(nm if self.infer_rg else 0): i for i, nm in enumerate(self.rgs)}

    def close(self):
        if getattr(self, 'peak', None) is not None:
            self.peak.close()
            self.peak = None
        if self.spool is not None:
            try:
                os.unlink(self.spool)
            except OSError:
                pass
            self.spool = None

    def _open_a(self, spool=True):
        sa = fastx.FastqStream(self.fastq[0])
        fd = None
        if not sa.regular and spool:
            import tempfile
            fd, self.spool = tempfile.mkstemp(prefix='kbbq-spool-', suffix='.fq', dir=os.environ.get('KBBQ_SPOOL_DIR') or None)
            sa.tee(fd)
        return sa, fd

    def pass1(self, minscore=6, tally=True, spool=True):
        """Count tables of the pair (None when no read was tallied); tally=False: file A alone, read groups and the spool
        only (a model file replaces pass 1); spool=False: no pass 2 will follow (fastq_to_covariate_arrays)."""
        sa, fd = self._open_a(spool)
        sb = fastx.FastqStream(self.fastq[1]) if tally else None
        budget = dev.device_budget()
        self.peak = _stream.Peak(budget)
        tables, b_ended = None, False
        segments = _prefetched(_pair_segments(sa, sb, _segment_bytes()))
        try:
            for a, b in segments:
                self.total_a += a.n
                if tally and b is None:
                    b_ended = True                               # file B ended at the last segment's end
                if b_ended:
                    if sa.regular or not spool:
                        break                                    # zip() has stopped (H6) and file A can be read again (or is not needed again)
                    continue                                     # ... a pipe is drained into the spool for pass 2
                with stage('scan'):
                    usable, S_seg, R, kind, idx = a.scan_next(b, self.infer_rg, self.rgs, self.longest)
                self.rgs = a.rg_names()
                self.longest = max(self.longest, S_seg)
                if tally and usable > 0:
                    tables = _grown(tables, R, 2 * self.longest)
                    try:
                        _stream.tally_range(a, b, self.infer_rg, a.first, a.first + usable, tables, minscore, budget)
                    except (IndexError, TypeError) as e:
                        if hasattr(e, 'read_index'):
                            e.read_index = a.first + max(e.read_index, 0)
                        raise
                self.usable += usable
                if kind:
                    raise fastx._SCAN_ERRORS[kind](a.first + idx)
                if tally and (b is None or b.n < a.n):
                    b_ended = True
        finally:
            segments.close()                                     # joins the reading thread before the streams go
            if fd is not None:
                os.close(fd)
            sa.close()
            if sb is not None:
                sb.close()
        return tables

    def pass2(self, lut, shape, output):
        """Every read of file A (its own first-appearance read groups, recalibrate.py:141-148) through K2 and the writer."""
        from . import _egress
        torch = dev._torch()
        device = torch.cuda.current_device()
        budget = dev.device_budget()
        sa = fastx.FastqStream(self.spool or self.fastq[0])
        peak = self.peak = getattr(self, 'peak', None) or _stream.Peak(budget)
        infer_rg = self.infer_rg
        segments = _prefetched(_pair_segments(sa, None, _segment_bytes()))

        def produced():
            rgs = []
            for a, _ in segments:
                with stage('scan'):
                    _, _, _, kind, idx = a.scan_next(None, infer_rg, rgs, 0)
                if kind:
                    raise fastx._SCAN_ERRORS[kind](a.first + idx)
                rgs = a.rg_names()
                try:
                    yield from _stream.produce_range(a, infer_rg, a.first, a.first + a.n, lut, shape, budget, origin=a.first,
                                                     extra=dict(text=a, base=a.first), device=device)
                except _stream.REFUSALS as e:
                    if hasattr(e, 'read_index'):
                        e.read_index = a.first + max(e.read_index, 0)
                    raise
        try:
            widest = fastx.pitch_for(max(self.longest, 1)) * 2 + 16
            if output is None:
                _egress.emit_produced(None, 0, produced(), widest)
            else:
                with open(output, 'wb') as sink:
                    _egress.emit_produced(None, 0, produced(), widest, sink=sink)
            LAST_RUN.clear()
            LAST_RUN['bands'] = []
            LAST_RUN['streamed'] = dict(peak.report(), reads=self.total_a, sequential=True, spooled=self.spool is not None,
                                        tables_bytes=8 * int(dev.N.load().kbbq_tables_count(shape[0], shape[2])), lut_bytes=int(lut.numel()))
        finally:
            try:
                segments.close()                                 # joins the reading thread before the stream goes
            except ValueError:                                   # (still running on the pipeline's thread: cannot happen once it has joined)
                pass
            sa.close()


def _recalibrate_sequential(fastq, infer_rg, gatkreport, output):
    """recalibrate_fastq for inputs that are read sequentially: same results, same errors at the same reads."""
    run = _Sequential(fastq, infer_rg)
    try:
        _warm_up()
        load = gatkreport is not None and os.path.exists(gatkreport)
        tables = run.pass1(tally=not load)
        if run.total_a == 0:
            return
        if load:
            tables = load_model(gatkreport, run.rg_to_int())
        elif tables is None:
            raise IndexError('index 0 is out of bounds for axis 0 with size 0')   # no read was tallied
        elif gatkreport is not None:
            save_model(tables, run.rg_to_int(), gatkreport)
        with stage('solve', sync=True):
            lut, shape = dev.solve_lut(tables)
        run.pass2(lut, shape, output)
    finally:
        run.close()


def _emit_streamed(text, single, lut, shape, output, world, rank):
    """Pass 2 of a shard that is not resident (kbbq/_stream.py): file A's reads are filled again slab by slab and go through
    K2 while the slabs before them are copied back, rendered and written -- fill | H2D | K2 | D2H | format | write.  The
    ranks do not agree on a first error beforehand as the resident pass 2 does: an error stops the rank that meets it where
    it stands, after the records before the offending slab have been written (the reference prints up to the offending read)."""
    from . import _egress
    st = single['streamed']
    torch = dev._torch()
    device = torch.cuda.current_device()
    peak = st.get('peak') or _stream.Peak(st['budget'])
    try:
        produced = _stream.produce_range(st['A'], st['infer_rg'], st['lo'], st['hi'], lut, shape, st['budget'], device=device)
        widest = fastx.pitch_for(single['S']) * 2 + 16
        if output is None:
            parallel.in_rank_order(lambda: _egress.emit_produced(text, single['first'], produced, widest))
        else:
            with open(output if world == 1 else '%s.rank%04d' % (output, rank), 'wb') as sink:
                _egress.emit_produced(text, single['first'], produced, widest, sink=sink)
        LAST_RUN.clear()
        LAST_RUN['bands'] = []
        LAST_RUN['streamed'] = dict(peak.report(), reads=st['hi'] - st['lo'],
                                    tables_bytes=8 * int(dev.N.load().kbbq_tables_count(shape[0], shape[2])), lut_bytes=int(lut.numel()))
    finally:
        peak.close()


def recalibrate_bam(bam, use_oq=False, set_oq=False):
    """Not implemented in the reference either."""
    raise NotImplementedError('Recalibrating a bam is not yet implemented. '
                              'Convert the BAM to FASTQ with `samtools fastq` first.')


def recalibrate(bam, fastq, infer_rg=False, use_oq=False, set_oq=False, gatkreport=None, output=None):
    """Dispatcher of `kbbq recalibrate` (reference recalibrate.py:166-174).  The reference raises NotImplementedError
    for ANY gatkreport; here `-g` works with FASTQ input (a deliberate, documented divergence: SURVEY.md 8(f) #3), and
    the cases that stay unimplemented -- no FASTQ input, or a BAM -- raise NotImplementedError as the reference does."""
    if gatkreport is not None and fastq is None:
        raise NotImplementedError('GATKreport reading / creation is only implemented for FASTQ input (-f).')
    if bam is not None:
        recalibrate_bam(bam, use_oq, set_oq)
    elif fastq is not None:
        with stage('[recalibrate_fastq, wall]'):
            recalibrate_fastq(fastq, infer_rg=infer_rg, gatkreport=gatkreport, output=output)
    else:
        raise ValueError('A BAM or FASTQ file should be provided for recalibration.')
