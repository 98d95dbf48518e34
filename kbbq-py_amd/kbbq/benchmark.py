"""
kbbq.benchmark -- truth-set calibration benchmark (reference kbbq/benchmark.py:9-164) with the
per-base work on the MI355X: the CIGAR walk that flags errors and skipped sites
(compare_reads.find_read_errors, K4) and the per-quality counting (calculate_q, K5).
Reference genome, variant sites and confident regions are read from TEXT formats
(kbbq/aln.py); the small per-genome set-up (skip mask) is host NumPy as in the reference.
"""
import numpy as np

from . import aln
from . import compare_reads
from . import fastx
from . import parallel


def get_ref_dict(reffilename):
    """{contig: array of 1-character strings} for every contig of the FASTA."""
    fasta = aln.FastaFile(reffilename)
    return {chrom: aln.chars(fasta.fetch(reference=chrom)) for chrom in fasta.references}


def get_var_sites(vcf):
    """{contig: [0-based positions covered by any record]}."""
    d = dict()
    for record in aln.read_vcf(vcf):
        d.setdefault(record.chrom, list()).extend(range(record.start, record.stop))
    return d


def get_bed_dict(refdict, bedfh):
    beddict = {chrom: np.zeros(len(refdict[chrom]), dtype=bool) for chrom in refdict.keys()}
    for rec in aln.read_bed(bedfh):
        beddict[rec.contig][rec.start:rec.end] = True
    return beddict


def get_full_skips(refdict, var_sites, bedfh=None):
    """Boolean skip mask per contig: variant sites, plus everything outside the BED if given."""
    skips = {chrom: np.zeros(len(refdict[chrom]), dtype=bool) for chrom in refdict.keys()}
    for chrom in skips.keys():
        skips[chrom][np.array(var_sites[chrom], dtype=np.int_)] = True
    if bedfh is not None:
        beddict = get_bed_dict(refdict, bedfh)
        for chrom in skips.keys():
            skips[chrom][~beddict[chrom]] = True
    return skips


def get_bam_readname(read):
    return read.query_name + ("/2" if read.is_read2 else "/1")


def get_fastq_readname(read):
    return read.name.split(sep='_')[0]


# ---------------------------------------------------------------------------
# device batch
# ---------------------------------------------------------------------------
class _Genome:
    """Concatenated contigs + skip mask on the device (uploaded once per benchmark call)."""

    def __init__(self, refdict, fullskips):
        from . import _device as dev
        torch = dev._torch()
        self.offset, parts, masks, pos = {}, [], [], 0
        for chrom, arr in refdict.items():
            self.offset[chrom] = pos
            parts.append(aln.codes(arr)); masks.append(np.asarray(fullskips[chrom], dtype=np.uint8))
            pos += len(arr)
        self.length = pos
        cat = lambda xs: np.concatenate(list(xs)) if xs else np.zeros(0, dtype=np.uint8)
        letters, flags = cat(parts), cat(masks)
        import os
        if os.environ.get('KBBQ_REFERENCE_MASK') != 'separate' and (not len(letters) or int(letters.max()) < 128):
            # ASCII reference: the skip flag rides in bit 7 of every byte (kbbq_find_errors_dev with a NULL mask) --
            # one scattered window per chunk instead of two
            self.genome = torch.from_numpy(np.ascontiguousarray(letters | ((flags != 0).astype(np.uint8) << 7))).cuda()
            self.mask = None
        else:
            self.genome = torch.from_numpy(np.ascontiguousarray(letters)).cuda()
            self.mask = torch.from_numpy(np.ascontiguousarray(flags)).cuda()
        self.sizes = {c: len(a) for c, a in refdict.items()}


def _read_arrays(reads, genome, flip_reverse, rows=None):
    """Host arrays K4 needs, from an aln.AlignmentFile (native reader: vectorised) or from a list of
    pysam-like read objects (one Python pass; the form the reference's own tests use).  rows = (lo, hi): only
    those alignments (one rank's shard); the row pitch is the whole file's, so shards agree on it."""
    if isinstance(reads, aln.AlignmentFile):
        b = reads.batch()
        lo, hi = (0, b.n) if rows is None else rows
        sl = slice(lo, hi)
        n = hi - lo
        lens = b.qlen[sl].astype(np.uint32)
        pitch = fastx.pitch_for(int(b.qlen.max()) if b.n else 1)
        seq = b.plane(0, pitch, lo, n)
        sizes = np.array([genome.sizes[c] for c in b.contig_names] + [0], dtype=np.int64)      # KeyError: unknown contig
        offs = np.array([genome.offset[c] for c in b.contig_names] + [0], dtype=np.int64)
        contig, pos = b.contig[sl], b.pos[sl]
        size = sizes[contig] if n else np.zeros(0, np.int64)
        start = np.minimum(np.maximum(pos, 0), size)                 # Python slice clamping of the reference window
        end = np.minimum(np.maximum(pos + b.ref_span[sl], start), size)
        ref_start = (offs[contig] if n else np.zeros(0, np.int64)) + start
        ref_len = (end - start).astype(np.int32)
        flip = ((b.flag[sl] & 16) != 0).astype(np.uint8) if flip_reverse else np.zeros(n, dtype=np.uint8)
        return n, lens, pitch, seq, ref_start, ref_len, b.cig_off[sl], b.cig_n[sl], b.cigar, flip   # offsets into the whole CIGAR array
    longest = max([len(r.query_sequence) for r in reads] + [1])
    if rows is not None:
        reads = reads[rows[0]:rows[1]]
    n = len(reads)
    lens = np.array([len(r.query_sequence) for r in reads], dtype=np.uint32)
    pitch = fastx.pitch_for(longest)
    seq = np.zeros((max(n, 1), pitch), dtype=np.uint8)
    ref_start = np.zeros(n, dtype=np.int64); ref_len = np.zeros(n, dtype=np.int32)
    cig_off = np.zeros(n, dtype=np.uint32); cig_n = np.zeros(n, dtype=np.uint32)
    flip = np.zeros(n, dtype=np.uint8)
    cigar = []
    for i, r in enumerate(reads):
        seq[i, :lens[i]] = aln.codes(r.query_sequence)
        size = genome.sizes[r.reference_name]                      # KeyError: unknown contig, as in the reference
        start = min(max(r.reference_start, 0), size)               # Python slice clamping of the reference window
        end = min(max(r.reference_end, start), size)
        ref_start[i] = genome.offset[r.reference_name] + start
        ref_len[i] = end - start
        cig_off[i] = len(cigar); cig_n[i] = len(r.cigartuples)
        for op, l in r.cigartuples:
            code = op if isinstance(op, (int, np.integer)) and 0 <= op <= 8 else 15   # 15: unrecognised -> ValueError
            cigar.append((int(l) << 4) | int(code))
        flip[i] = 1 if (flip_reverse and r.is_reverse) else 0
    return n, lens, pitch, seq, ref_start, ref_len, cig_off, cig_n, np.array(cigar, dtype=np.uint32), flip


def _upload_reads(reads, genome, flip_reverse, rows=None):
    """The arrays K4 reads, on the device: dict(n, lens (host), pitch, seq, len, cigar, ref_start, ref_len, flip, cig_off, cig_n)
    of an aln.AlignmentFile or a list of read objects; rows = (lo, hi): only those alignments (one rank's shard)."""
    from . import _device as dev
    torch = dev._torch()
    n, lens, pitch, seq, ref_start, ref_len, cig_off, cig_n, cigar, flip = _read_arrays(reads, genome, flip_reverse, rows)
    pad = lambda a, dt: np.ascontiguousarray(a if len(a) else np.zeros(1), dtype=dt)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return dict(n=n, lens=lens, pitch=pitch, seq=up(seq), len=up(pad(lens, np.uint32).view(np.int32)),
                cigar=up(pad(cigar, np.uint32).view(np.int32)), ref_start=up(pad(ref_start, np.int64)),
                ref_len=up(pad(ref_len, np.int32)), flip=up(pad(flip, np.uint8)),
                cig_off=up(pad(cig_off, np.uint32).view(np.int32)), cig_n=up(pad(cig_n, np.uint32).view(np.int32)))


def _find_errors(u, genome, fused=False):
    """K4 over uploaded reads (_upload_reads) -> (err, skip) device planes [n, pitch]; fused: ONE plane of flags (bit 0 error,
    bit 1 skip) comes back as `err` and `skip` is None -- the form K5 / K6 read when they are the only consumers."""
    from . import _device as dev
    from . import _native as N
    torch = dev._torch()
    n, pitch = u['n'], u['pitch']
    err = torch.zeros((max(n, 1), pitch), dtype=torch.uint8, device='cuda')
    skip = None if fused else torch.zeros((max(n, 1), pitch), dtype=torch.uint8, device='cuda')
    ctx = dev.context()
    N.check(N.load().kbbq_find_errors_dev(ctx.handle, N.ptr(u['seq']), N.ptr(u['len']), n, pitch,
                                          N.ptr(u['ref_start']), N.ptr(u['ref_len']), N.ptr(u['cig_off']), N.ptr(u['cig_n']),
                                          N.ptr(u['cigar']), N.ptr(genome.genome), N.ptr(genome.mask), genome.length,
                                          N.ptr(u['flip']), N.ptr(err), N.ptr(skip)))
    ctx.status()
    return err, skip


def _flag_batch(reads, genome, flip_reverse, keep=None, rows=None, fused=False):
    """K4 over aligned reads (an aln.AlignmentFile or a list of read objects) -> (err, skip) device planes
    [n, pitch], lens (host).  `keep`, a dict, receives the device seq plane for callers that go on to K6.
    rows = (lo, hi): only those alignments (one rank's shard).  fused: see _find_errors."""
    u = _upload_reads(reads, genome, flip_reverse, rows)
    if keep is not None:
        keep['seq'] = u['seq']
    err, skip = _find_errors(u, genome, fused)
    return err, skip, u['lens'], u['pitch']


def _match_native(bam, fq):
    """Row of the alignment for every FASTQ read (benchmark.py:102-124), matched in C++."""
    from . import _native as N
    n = fq.n
    idx = np.zeros(max(n, 1), dtype=np.int64)
    N.check(N.load().kbbq_sam_match_fastq(bam.batch()._handle.h, fq._h, N.ptr(idx)))
    idx = idx[:n]
    if n and int(idx.min()) < 0:
        raise KeyError(fq.name(int(np.flatnonzero(idx < 0)[0])).split('_')[0])
    return idx


def _bam_names(reads):
    """Canonical read names (benchmark.py:41-48): QNAME + /1 or /2."""
    if isinstance(reads, aln.AlignmentFile):
        b = reads.batch()
        second = (b.flag & 128) != 0
        return [nm + ('/2' if s else '/1') for nm, s in zip(b.names(), second)]
    return [get_bam_readname(r) for r in reads]


def _count_q(qual, err, skip, lens, pitch, qoffset, reduce=False):
    """K5 -> (numerrs[256], numtotal[256]) as int64 host arrays; reduce: summed over all ranks."""
    from . import _device as dev
    from . import _native as N
    torch = dev._torch()
    n = len(lens)
    counts = torch.zeros(512, dtype=torch.int64, device='cuda')
    d_len = torch.from_numpy(np.ascontiguousarray(lens.astype(np.uint32)).view(np.int32)).cuda() if n else torch.zeros(1, dtype=torch.int32, device='cuda')
    ctx = dev.context()
    N.check(N.load().kbbq_count_q_dev(ctx.handle, N.ptr(qual), N.ptr(err), N.ptr(skip), N.ptr(d_len),
                                      n, pitch, qoffset, N.ptr(counts)))
    ctx.status()
    if reduce:
        parallel.allreduce_tables(counts)                        # one 4 KB sum over the ranks' shards
    h = counts.cpu().numpy()
    return h[256:].copy(), h[:256].copy()


def _actual_q(numerrs, numtotal):
    """benchmark.py:83-91 from the two count vectors (trimmed to the largest observed quality)."""
    top = np.flatnonzero(numtotal)
    size = int(top[-1]) + 1 if top.size else 0
    numtotal, numerrs = numtotal[:size], numerrs[:size]
    nonzero = numtotal != 0
    q = compare_reads.p_to_q(np.true_divide(numerrs[nonzero], numtotal[nonzero]))
    actual_q = np.zeros(len(numtotal), dtype=np.int_)
    actual_q[nonzero] = q
    return actual_q, numtotal.astype(np.int_)


def get_error_dict(bamfile, refdict, fullskips):
    """{canonical read name: (errors, skips)} -- flags of reverse-strand reads flipped, because a
    FASTQ made from the BAM holds them reverse-complemented."""
    reads = bamfile if isinstance(bamfile, aln.AlignmentFile) else list(bamfile)
    genome = _Genome(refdict, fullskips)
    err, skip, lens, _ = _flag_batch(reads, genome, flip_reverse=True)
    e, s = err.cpu().numpy().astype(bool), skip.cpu().numpy().astype(bool)
    return {name: (e[i, :lens[i]].copy(), s[i, :lens[i]].copy()) for i, name in enumerate(_bam_names(reads))}


def calculate_q(errors, quals):
    """(actual_q, numtotal) indexed by predicted quality, from flat arrays (reference
    benchmark.py:76-91).  The two bincounts run on the device (K5)."""
    from . import _device as dev
    torch = dev._torch()
    errors = np.asarray(errors, dtype=bool).reshape(-1)
    quals = np.asarray(quals).reshape(-1)
    if quals.size and (quals.min() < 0 or quals.max() > 255):
        raise ValueError('qualities must lie in 0..255')
    n = quals.size
    pitch = fastx.pitch_for(max(n, 1))
    if pitch > 65536:                      # long flat input: fold into rows of 4096
        pitch = 4096
    rows = max(1, -(-n // pitch))
    q = np.zeros(rows * pitch, dtype=np.uint8); q[:n] = quals
    e = np.zeros(rows * pitch, dtype=np.uint8); e[:n] = errors
    lens = np.full(rows, pitch, dtype=np.uint32); lens[-1] = n - (rows - 1) * pitch
    up = lambda a: torch.from_numpy(a.reshape(rows, pitch)).cuda()
    d_q, d_e = up(q), up(e)
    d_s = torch.zeros((rows, pitch), dtype=torch.uint8, device='cuda')
    numerrs, numtotal = _count_q(d_q, d_e, d_s, lens, pitch, 0)
    return _actual_q(numerrs, numtotal)


def get_bamread_quals(read, use_oq=False):
    if use_oq:
        return np.array([ord(c) - 33 for c in read.get_tag('OQ')], dtype=np.int_)
    return np.array(read.query_qualities, dtype=np.int_)


def _qual_plane(reads, lens, pitch, use_oq, rows=None):
    """uint8 plane of phred values (not characters) per read (of alignments rows = (lo, hi) only)."""
    if isinstance(reads, aln.AlignmentFile):
        b = reads.batch()
        lo, hi = (0, b.n) if rows is None else rows
        n = hi - lo
        have = (b.oq_len if use_oq else b.qual_len)[lo:hi]
        if use_oq and n and int(have.min()) < 0:
            raise KeyError("tag 'OQ' not present")
        if n and np.any(have != lens):
            i = int(np.flatnonzero(have != lens)[0])
            raise _at(IndexError('boolean index did not match indexed array: read %d has %d qualities for %d bases'
                                 % (lo + i, int(have[i]), int(lens[i]))), i)
        chars = b.plane(2 if use_oq else 1, pitch, lo, n)
        inside = np.arange(pitch)[None, :] < np.asarray(lens)[:, None] if n else np.zeros((1, pitch), dtype=bool)
        if n and int(chars[:n][inside].min(initial=255)) < 33:
            raise ValueError('qualities must lie in 0..255')
        return np.where(inside, chars[:max(n, 1)] - 33, 0).astype(np.uint8)
    if rows is not None:
        reads = reads[rows[0]:rows[1]]
    q = np.zeros((max(len(reads), 1), pitch), dtype=np.uint8)
    for i, r in enumerate(reads):
        v = get_bamread_quals(r, use_oq)
        if len(v) != lens[i]:
            raise IndexError('boolean index did not match indexed array: read %d has %d qualities for %d bases'
                             % (i, len(v), lens[i]))
        if v.size and (v.min() < 0 or v.max() > 255):
            raise ValueError('qualities must lie in 0..255')
        q[i, :lens[i]] = v
    return q


def _qual_chars_dev(reads, lens, pitch, use_oq, rows=None):
    """The quality characters (QUAL, or the OQ tag) of a native reader's alignments as a device plane [n, pitch], zero behind every
    read -- checked like _qual_plane: lengths on the host arrays, the value range on the device (every byte below '!' must be padding)."""
    from . import _device as dev
    torch = dev._torch()
    b = reads.batch()
    lo, hi = (0, b.n) if rows is None else rows
    n = hi - lo
    have = (b.oq_len if use_oq else b.qual_len)[lo:hi]
    if use_oq and n and int(have.min()) < 0:
        raise KeyError("tag 'OQ' not present")
    if n and np.any(have != lens):
        i = int(np.flatnonzero(have != lens)[0])
        raise _at(IndexError('boolean index did not match indexed array: read %d has %d qualities for %d bases'
                             % (lo + i, int(have[i]), int(lens[i]))), i)
    chars = torch.from_numpy(b.plane(2 if use_oq else 1, pitch, lo, n)).cuda()
    if n and int((chars[:n] < 33).sum().item()) != n * pitch - int(np.asarray(lens, dtype=np.int64).sum()):
        raise ValueError('qualities must lie in 0..255')
    return chars


def _at(exc, index):
    """Tag an exception with the read of this rank's shard it is about (as the kernels' status does): the ranks then
    agree on the first one (parallel.raise_first_error)."""
    exc.read_index = int(index)
    return exc


def _on_all_ranks(fn, first=0):
    """fn() on this rank's shard; an exception of any rank's data is raised by every rank (the one of the smallest
    read index), so that nobody is left waiting in the sum of the counts."""
    exc, res = None, None
    try:
        res = fn()
    except (IndexError, TypeError, ValueError, KeyError) as e:
        exc = e
    parallel.raise_first_error(exc, None if exc is None else first + max(getattr(exc, 'read_index', 0), 0))
    return res


def benchmark_bam(bamfile, ref, var_sites, use_oq=False, bedfh=None):
    """Under torch.distributed (parallel.init_from_env) every rank flags and counts a contiguous shard of the
    alignments; the per-quality counts are summed with one allreduce (SURVEY 8(e) applied to this path)."""
    from . import _device as dev
    torch = dev._torch()
    fullskips = get_full_skips(ref, var_sites, bedfh)
    reads = bamfile if isinstance(bamfile, aln.AlignmentFile) else list(bamfile)
    world, rank = parallel.world_rank()
    rows = parallel.shard_range(len(reads), rank, world) if world > 1 else None
    genome = _Genome(ref, fullskips)

    def shard():
        err, skip, lens, pitch = _flag_batch(reads, genome, flip_reverse=False, rows=rows, fused=True)
        if isinstance(reads, aln.AlignmentFile):
            # the native reader's plane of quality CHARACTERS goes up as it is and K5 subtracts the 33 (three masked NumPy passes over
            # the plane were 60 % of this function: 47 ms per 200 K alignments); a character below '!' is the reference's ValueError
            qual, offset = _qual_chars_dev(reads, lens, pitch, use_oq, rows), 33
        else:
            qual, offset = torch.from_numpy(_qual_plane(reads, lens, pitch, use_oq, rows)).cuda(), 0
        return qual, err, skip, lens, pitch, offset
    qual, err, skip, lens, pitch, offset = _on_all_ranks(shard, rows[0] if rows else 0)
    return _actual_q(*_count_q(qual, err, skip, lens, pitch, offset, reduce=world > 1))


LAST_RUN = {}            # what the most recent benchmark_fastq flagged on this rank (tests, KBBQ_TIMING): K4's share of the alignments


def benchmark_fastq(fqfile, bamfile, ref, var_sites, bedfh=None):
    """Under torch.distributed every rank counts a contiguous shard of the FASTQ reads and flags (K4) ONLY the
    alignments its reads map to: the name join of the reference (benchmark.py:93-104) is one host pass every rank makes
    over the names -- no flags, no bases -- and tells the rank the span of alignment rows [first, last] its shard needs
    (a FASTQ made from the BAM keeps its order: about 1 / world of the rows); K4 runs over that span, K5 over the shard,
    and the 512 per-quality counters are summed with one allreduce."""
    from . import _device as dev
    import sys
    torch = dev._torch()
    fullskips = get_full_skips(ref, var_sites, bedfh)
    reads = bamfile if isinstance(bamfile, aln.AlignmentFile) else list(bamfile)
    fq = fastx.NativeFastq(fqfile)
    if isinstance(reads, aln.AlignmentFile):
        idx = _match_native(reads, fq)                               # both sides native: no Python object per read
    else:
        row = {}
        for i, name in enumerate(_bam_names(reads)):
            row[name] = i                                            # later reads replace earlier ones (dict)
        idx = np.array([row[fq.name(i).split('_')[0]] for i in range(fq.n)], dtype=np.int64)   # KeyError if absent
    n, S, _, kind, bad = fq.scan(None, False)
    world, rank = parallel.world_rank()
    lo, hi = parallel.shard_range(fq.n, rank, world) if world > 1 else (0, fq.n)
    m = hi - lo
    idx = idx[lo:hi]
    nalign = reads.batch().n if isinstance(reads, aln.AlignmentFile) else len(reads)
    # the rows K4 walks on this rank: what its FASTQ reads map to, and its own contiguous share of the alignments (so that the
    # ranks together look at EVERY alignment, as get_error_dict does -- bad input is reported whoever's reads map to it)
    a_lo, a_hi = parallel.shard_range(nalign, rank, world) if world > 1 else (0, nalign)
    if m:
        a_lo, a_hi = min(a_lo, int(idx.min())), max(a_hi, int(idx.max()) + 1)
    genome = _Genome(ref, fullskips)

    def flag_shard():
        return _flag_batch(reads, genome, flip_reverse=True, fused=True, rows=(a_lo, a_hi))
    err, skip, lens, pitch = _on_all_ranks(flag_shard, a_lo)
    LAST_RUN.clear()
    LAST_RUN.update(rank=rank, world=world, fastq_reads=m, alignments=nalign, k4_alignments=a_hi - a_lo)
    from . import _trace
    if _trace.ON:
        sys.stderr.write('kbbq benchmark: rank %d of %d counts FASTQ reads [%d, %d) and flagged alignments [%d, %d): %d of %d\n'
                         % (rank, world, lo, hi, a_lo, a_hi, a_hi - a_lo, nalign))
    idx = idx - a_lo
    _, _, fqual, fmeta = fq.fill(None, False, m, max(pitch, fastx.pitch_for(S)), first=lo)
    flens = (fmeta & 0xFFFF).astype(np.uint32)

    def check():
        if np.any(flens != lens[idx]):
            raise _at(IndexError('boolean index did not match indexed array: FASTQ and BAM read lengths differ'),
                      int(np.flatnonzero(flens != lens[idx])[0]))
    _on_all_ranks(check, lo)
    d_idx = torch.from_numpy(np.ascontiguousarray(idx)).cuda()
    fp = fqual.shape[1]
    e = torch.zeros((max(m, 1), fp), dtype=torch.uint8, device='cuda')          # the flags of every FASTQ read's alignment
    if m:
        e[:m, :pitch] = err.index_select(0, d_idx)
    return _actual_q(*_count_q(torch.from_numpy(fqual).cuda(), e, None, flens, fp, 33, reduce=world > 1))


def print_benchmark(actual_q, label, nbases):
    """Tab-separated rows: predicted quality, actual quality, label, number of bases."""
    nonzero = (nbases != 0)
    for pq, aq, nb in zip(np.arange(len(actual_q))[nonzero], actual_q[nonzero], nbases[nonzero]):
        print(pq, aq, label, nb, sep="\t")


def benchmark(bamfile, fafile, vcffile, fastqfile=None, label=None, use_oq=False, bedfh=None):
    """Run the benchmark and print it.  With a FASTQ, its reads are matched to the alignments by name."""
    bam = aln.AlignmentFile(bamfile, 'r')
    ref = get_ref_dict(fafile)
    var_sites = get_var_sites(vcffile)
    if fastqfile is not None:
        actual_q, nbases = benchmark_fastq(fastqfile, bam, ref, var_sites, bedfh)
        label = (fastqfile if label is None else label)
    else:
        actual_q, nbases = benchmark_bam(bam, ref, var_sites, use_oq, bedfh)
        label = (bamfile if label is None else label)
    if parallel.world_rank()[1] == 0:                                # every rank holds the same totals
        print_benchmark(actual_q, label, nbases)
