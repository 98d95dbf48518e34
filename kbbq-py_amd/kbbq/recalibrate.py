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
                                 gather=parallel.all_gather_object if world > 1 else None)
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
        if packed.get('other') is not None:
            fastx.close_later(packed.pop('other'))     # the corrected file is not needed past the tally: unmap it off the critical path
    if tables is not None and packed.get('total', packed['n']) == 0:
        tables = None
    return packed, tables


def fastq_to_covariate_arrays(fastq, infer_rg=False, minscore=6, maxscore=42):
    """Tally errors and observations of the (uncorrected, corrected) FASTQ pair by read
    group, reported quality, cycle and dinucleotide.  Returns the reference's 9-tuple:
    meanq, rg_errs, rg_total, q_errs, q_total, pos_errs, pos_total, dinuc_errs, dinuc_total."""
    packed, tables = _pack_and_tally(fastq, infer_rg, minscore, maxscore)
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
    shard = (rank, world) if world > 1 else None
    packed, single = None, None
    if gatkreport is not None and os.path.exists(gatkreport):
        scan = fastx.PairScan(fastq[0], None, infer_rg)
        _warm_up()
        text = scan.result()[0]
        if text.n == 0:
            return
        single = fastx.pack_single(text, infer_rg, shard, bands=True, to_device=True)
        tables = load_model(gatkreport, single['rg_to_int'])
    else:
        packed, tables = _pack_and_tally(fastq, infer_rg, 6, 42)
        text = packed['text']
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
            single = fastx.pack_single(text, infer_rg, shard, bands=True, to_device=True)

    if single['R'] != R:
        # pass 2 met read groups the model does not have (recalibrate.py:143-151: an IndexError at the first such read) or
        # fewer than it has: rows gathered by THEIR read-group segments do not fit the model's -- one read per row decides
        for band in single['bands']:
            band['laid'] = None

    def apply_band(band):
        """New qualities of a band: in the band's own layout (mate-pair rows stay mate-pair rows, stored in input order
        -- the writer reads them as they are) or, when the layout's kernel cannot serve the LUT or the rows, one read
        per row from the checked kernel."""
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
