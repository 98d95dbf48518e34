"""
kbbq.compare_reads -- the hot-path subset of the reference module of the same
name (reference kbbq/compare_reads.py:141-328), MI355X-backed.

Kept names: RescaledNormal, Dinucleotide, gatk_delta_q, p_to_q, q_to_p,
generic_cycle_covariate, generic_dinuc_covariate, fastq_cycle_covariates,
fastq_dinuc_covariates, fastq_infer_secondinpair, fastq_infer_rg and the
per-read apply recalibrate_fastq.  The BAM/VCF truth-set helpers and the
regression experiment of the reference file are out of scope.

What runs where: the per-read apply goes through the K2 HIP kernel (C ABI
kbbq_apply) and gatk_delta_q through the K3 kernel (kbbq_delta_q_dev, with the
transcendental terms from the host's SciPy, kbbq/_solve.py); p_to_q / q_to_p and
the prior table are host NumPy with the reference's own calls (SURVEY hazards
H4/H5); the small covariate helpers are plain NumPy views of the same
definitions the kernels implement.
"""
import numpy as np

from . import _native as N

_MAXSCORE = 42


def find_read_errors(read, ref, variable):
    """(errors, skips) of one aligned read from its CIGAR (reference compare_reads.py:84-139):
    M/=/X compare with the reference and take the site mask, insertions are skipped when both
    flanking sites are, deletions / N mark the previous base when they cover a masked site, soft
    clips are skipped; indels are not counted as errors.  Runs on the device (K4)."""
    from . import benchmark as _bm
    chrom = read.reference_name
    g = _bm._Genome({chrom: ref[chrom]}, {chrom: variable[chrom]})
    for op, _ in read.cigartuples:
        if not (isinstance(op, (int, np.integer)) and 0 <= op <= 8):
            raise ValueError("Unrecognized Cigar Operation " + str(op) + " In Read\n" + str(read))
    err, skip, lens, _ = _bm._flag_batch([read], g, flip_reverse=False)
    L = int(lens[0])
    return err[0, :L].cpu().numpy().astype(bool), skip[0, :L].cpu().numpy().astype(bool)


class RescaledNormal:
    """Cached log-prior of the Bayesian delta-Q model: a normal density over the
    quality-score difference d, sigma = 0.5, rescaled to 0.9 at d = 0
    (reference compare_reads.py:141-191).  ``prior_dist[d]`` is longdouble and
    -inf once exp() underflows (d >= 19), exactly as the reference's trapped
    FloatingPointError makes it."""

    maxscore = _MAXSCORE
    possible_diffs = np.arange(_MAXSCORE + 1, dtype=np.int_)

    @staticmethod
    def _table(n):
        tiny = np.finfo(np.float64).tiny
        tab = np.empty(n, dtype=np.longdouble)
        with np.errstate(under='ignore', divide='ignore'):
            for d in range(n):
                dens = np.exp(-((np.int_(d) / .5) ** 2) / 2)
                # the reference evaluates under np.seterr(all='raise'): a subnormal exp()
                # result traps and the entry becomes -inf (compare_reads.py:166-180)
                tab[d] = np.log(.9 * dens) if dens >= tiny else -np.inf
        return tab

    prior_dist = _table.__func__(_MAXSCORE + 1)

    @classmethod
    def prior(cls, difference):
        return cls.prior_dist[difference]


class Dinucleotide:
    """Nucleotide order A, T, G, C; dinucleotide index = 4 * first + second
    (reference compare_reads.py:193-233)."""

    nucleotides = ['A', 'T', 'G', 'C']
    complement = {'A': 'T', 'T': 'A', 'G': 'C', 'C': 'G'}
    dinucs = [a + b for a in ['A', 'T', 'G', 'C'] for b in ['A', 'T', 'G', 'C']]
    dinuc_to_int = {d: i for i, d in enumerate(dinucs)}

    @classmethod
    def vecget(cls, dinucs, *args, **kwargs):
        arr = np.asarray(dinucs)
        flat = [cls.dinuc_to_int.get(str(x), *args) for x in arr.ravel()]
        if any(v is None for v in flat):
            raise TypeError("int() argument must be a string, a bytes-like object or a real number, not 'NoneType'")
        return np.array(flat, dtype=np.int_).reshape(arr.shape)

    @classmethod
    def veccomplement(cls, nucs, *args, **kwargs):
        arr = np.asarray(nucs)
        flat = [cls.complement.get(str(x), *args) for x in arr.ravel()]
        return np.array(flat, dtype=np.str_).reshape(arr.shape)


def q_to_p(q):
    """10^(-q/10) in float64, widened to longdouble (reference compare_reads.py:269-271)."""
    q = np.asarray(q)
    return np.array(np.power(10.0, -(q / 10.0)), dtype=np.longdouble, copy=True)


def p_to_q(p, maxscore=_MAXSCORE):
    """Truncated -10 log10(p); p == 0 -> maxscore; clipped to [0, maxscore]
    (reference compare_reads.py:262-267)."""
    p = np.asarray(p)
    q = np.zeros(p.shape, dtype=np.int_)
    nz = p != 0
    q[nz] = (-10.0 * np.log10(p[nz])).astype(np.int_)
    q[~nz] = maxscore
    return np.clip(q, 0, maxscore).copy()


def gatk_delta_q(prior_q, numerrs, numtotal, maxscore=_MAXSCORE):
    """MAP quality minus prior quality for every cell (reference compare_reads.py:235-260):
    argmax over q' = 0..42 of  prior_dist[|q' - prior_q|] + logpmf(errs+1; total+2, 10^(-q'/10)),
    first maximum wins.  Runs on the device (K3 kernel, C ABI kbbq_delta_q_dev); the host
    supplies the gammaln terms with the SciPy functions logpmf itself uses (kbbq._solve)."""
    prior_q = np.asarray(prior_q)
    assert prior_q.shape == np.shape(numerrs) == np.shape(numtotal)
    if maxscore != _MAXSCORE:
        raise ValueError('the prior table and the device solve are fixed at maxscore = 42')
    from ._device import delta_q
    return delta_q(prior_q, numerrs, numtotal)


def bamread_get_oq(read):
    """Original qualities from the OQ tag, phred+33 text -> int array (reference
    compare_reads.py:332-336)."""
    return np.frombuffer(read.get_tag('OQ').encode('utf-32-le'), dtype=np.uint32).astype(np.int_) - 33


def get_rg_to_pu(bamfileobj):
    """{read group ID: platform unit} in header order (reference compare_reads.py:338-340)."""
    return {rg['ID']: rg['PU'] for rg in bamfileobj.header.as_dict()['RG']}


def generic_cycle_covariate(sequencelen, secondinpair=False):
    """0..L-1, or -1..-L for second-in-pair reads (reference compare_reads.py:275-279)."""
    cycle = np.arange(sequencelen)
    return np.negative(cycle + 1) if secondinpair else cycle


_CODE = np.full(256, -1, dtype=np.int_)
for _i, _c in enumerate('ATGC'):
    _CODE[ord(_c)] = _i


def generic_dinuc_covariate(sequences, quals, minscore=6):
    """Dinucleotide context per base (reference compare_reads.py:281-293): -1 at position 0,
    where q < minscore, and where either base is N; otherwise 4 * code(prev) + code(cur)."""
    sequences = np.asarray(sequences)
    quals = np.asarray(quals)
    assert sequences.shape == quals.shape
    assert sequences.dtype == np.dtype('U1')
    codes = sequences.view(np.uint32).reshape(sequences.shape)
    out = np.zeros(sequences.shape, dtype=np.int_)
    out[..., 0] = -1
    isn = sequences == 'N'
    invalid = (quals[..., 1:] < minscore) | isn[..., 1:] | isn[..., :-1]
    small = np.where(codes < 256, codes, 0)
    c = _CODE[small]
    c[codes >= 256] = -1
    if np.any(((c[..., 1:] < 0) | (c[..., :-1] < 0)) & ~invalid):
        raise TypeError("int() argument must be a string, a bytes-like object or a real number, not 'NoneType'")
    out[..., 1:] = np.where(invalid, -1, 4 * c[..., :-1] + c[..., 1:])
    return out


def fastq_cycle_covariates(read, secondinpair=False):
    return generic_cycle_covariate(len(read.sequence), secondinpair)


def fastq_dinuc_covariates(read, minscore=6):
    quals = np.array(read.get_quality_array(), dtype=np.int_)
    return generic_dinuc_covariate(np.array(list(read.sequence), dtype='U1'), quals, minscore)


def fastq_infer_secondinpair(read):
    """True when the first '_' field of the name ends in '/2' (reference compare_reads.py:304-306)."""
    return read.name.split(sep='_')[0][-2:] == '/2'


def fastq_infer_rg(read):
    """Read group from a 'name_RG:Z:id' style name (reference compare_reads.py:308-318)."""
    rgstr = read.name.split(sep='_')[1]
    assert rgstr[0:2] == 'RG'
    return rgstr.split(':')[-1]


def recalibrate_fastq(read, meanq, globaldeltaq, qscoredeltaq, positiondeltaq, dinucdeltaq, rg,
                      dinuc_to_int, secondinpair=False, minscore=6, maxscore=_MAXSCORE):
    """Recalibrated qualities of ONE read (reference compare_reads.py:320-328), computed by
    the K2 HIP kernel through the host-buffer C entry point kbbq_apply."""
    from ._device import context
    rg = int(np.asarray(rg).reshape(-1)[0])
    seq = read.sequence.encode('latin-1')
    q = np.array(read.get_quality_array(), dtype=np.int64)
    L = len(seq)
    if np.any(q + 33 < 0) or np.any(q + 33 > 255):
        raise ValueError('quality outside the byte range')
    pitch = max(16, (L + 15) // 16 * 16)
    sp = np.full((1, pitch), ord('N'), dtype=np.uint8)
    qp = np.zeros((1, pitch), dtype=np.uint8)
    sp[0, :L] = np.frombuffer(seq, dtype=np.uint8)
    qp[0, :L] = (q + 33).astype(np.uint8)
    meta = np.array([L | (rg << 16) | (int(bool(secondinpair)) << 31)], dtype=np.uint32)
    a = [np.ascontiguousarray(np.asarray(x), dtype=np.int64)
         for x in (meanq, globaldeltaq, qscoredeltaq, positiondeltaq, dinucdeltaq)]
    R, Qt, S2 = a[3].shape
    D = a[4].shape[2]
    out = np.zeros((1, pitch), dtype=np.uint8)
    ctx = context()
    N.check(N.load().kbbq_apply(ctx.handle, N.ptr(sp), N.ptr(qp), N.ptr(meta), 1, pitch, R, Qt, S2, D,
                                minscore, N.ptr(a[0]), N.ptr(a[1]), N.ptr(a[2]), N.ptr(a[3]), N.ptr(a[4]),
                                N.ptr(out)))
    return out[0, :L].astype(np.int_) - 33


def tstamp():
    """'[ YYYY-MM-DD HH:MM:SS ]' (reference compare_reads.py:26-33)."""
    import datetime
    return '[ ' + datetime.datetime.today().isoformat(' ', 'seconds') + ' ]'


def load_positions(posfile):
    """{contig: [0-based positions]} covered by the lines of an uncompressed BED file (reference
    compare_reads.py:35-52)."""
    d = dict()
    with open(posfile, 'r') as infh:
        for line in infh:
            chrom, pos, end = line.rstrip().split()
            d.setdefault(chrom, list()).extend(range(int(pos), int(end)))
    return d


def get_var_sites(vcf):
    """{contig: [0-based positions covered by any record]} of a VCF file (reference compare_reads.py:54-68; the
    same function as kbbq.benchmark's)."""
    from . import benchmark
    return benchmark.get_var_sites(vcf)

