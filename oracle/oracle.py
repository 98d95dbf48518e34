"""
oracle.py -- CPU ORACLE for the kbbq recalibrate hot path.  TEST INFRASTRUCTURE ONLY.

This module (with kbbq_oracle.c next to it) restates, on the CPU, the algorithm
of the reference (adamjorr/kbbq-py @ v1) for the path BASELINE.json names:
error flagging -> covariate binning -> delta-Q solve -> apply.  It is the
CHECKER of the HIP path.  Only tests/, __graft_entry__.smoke() and bench.py's
``cpu_baseline`` leg may import it; nothing under kbbq-py_amd/ does.

Parity pinning (DESIGN.md section "Oracle"):
  * restated known answers of the reference's own tests
    (tests/test_recalibrate.py:19-99, tests/test_compare_reads.py:124-233,
    tests/test_gatk_applybqsr.py:105-121) -> tests/test_oracle_known_answers.py
  * golden vectors produced by running the UNMODIFIED reference in the build
    container under oracle/_shim.py (oracle/gen_golden.py -> tests/golden/)
    -> tests/test_oracle_golden.py

Citations are relative to /root/reference/.
"""
import ctypes
import hashlib
import os
import subprocess

import numpy as np
import scipy.stats

HERE = os.path.dirname(os.path.abspath(__file__))
MAXSCORE = 42
MINSCORE = 6

# floor(2^32 * 10^(-q/10)), q = 0..42, thr[0] clamped to 2^32-1 (synthetic
# error thresholds); computed once with 60-digit decimals.
SYNTH_THR = np.array([
    4294967295, 3411613790, 2709941159, 2152582777, 1709857277, 1358187913,
    1078847007, 856958638, 680706442, 540704347, 429496729, 341161379,
    270994115, 215258277, 170985727, 135818791, 107884700, 85695863, 68070644,
    54070434, 42949672, 34116137, 27099411, 21525827, 17098572, 13581879,
    10788470, 8569586, 6807064, 5407043, 4294967, 3411613, 2709941, 2152582,
    1709857, 1358187, 1078847, 856958, 680706, 540704, 429496, 341161, 270994],
    dtype=np.uint32)

KO_INDEX_ERROR = -2
KO_TYPE_ERROR = -3


# --------------------------------------------------------------------------
# model numerics (pure numpy / scipy, same library calls as the reference)
# --------------------------------------------------------------------------
def q_to_p(q):
    """compare_reads.py:269-271 -- float64 power, then widened to longdouble."""
    q = np.asarray(q)
    return np.array(np.power(10.0, -(q / 10.0)), dtype=np.longdouble, copy=True)


def p_to_q(p, maxscore=MAXSCORE):
    """compare_reads.py:262-267 -- TRUNCATING -10*log10(p); p == 0 -> maxscore; clip."""
    p = np.asarray(p)
    q = np.zeros(p.shape, dtype=np.int64)
    nz = p != 0
    q[nz] = (-10.0 * np.log10(p[nz])).astype(np.int64)
    q[~nz] = maxscore
    return np.clip(q, 0, maxscore).copy()


def _prior_dist(maxscore=MAXSCORE):
    """compare_reads.py:166-180 RescaledNormal.prior_dist: log(.9*exp(-((d/.5)**2)/2))
    evaluated in float64 under np.seterr(all='raise'); an underflowing exp
    becomes -inf.  Stored as longdouble."""
    out = np.zeros(maxscore + 1, dtype=np.longdouble)
    diffs = np.arange(maxscore + 1, dtype=np.int_)
    old = np.seterr(all='raise')
    try:
        for i in range(diffs.shape[0]):
            try:
                out[i] = np.log(.9 * np.exp(-((diffs[i] / .5) ** 2) / 2))
            except FloatingPointError:
                out[i] = -np.inf
    finally:
        np.seterr(**old)
    return out


PRIOR_DIST = _prior_dist()


def gatk_delta_q(prior_q, numerrs, numtotal, maxscore=MAXSCORE):
    """compare_reads.py:235-260.  argmax over q' in 0..maxscore of
    prior_dist[|q'-prior_q|] (longdouble) + binom.logpmf(errs+1, total+2, 10^(-q'/10))
    (float64), first maximum wins; returns argmax - prior_q."""
    prior_q = np.asarray(prior_q)
    numerrs = np.asarray(numerrs)
    numtotal = np.asarray(numtotal)
    assert prior_q.shape == numerrs.shape == numtotal.shape
    possible_q = np.arange(maxscore + 1, dtype=np.int64)
    diff = np.absolute(np.subtract.outer(possible_q, prior_q).astype(np.int64))
    prior = PRIOR_DIST[diff]
    b_errs = np.broadcast_to(numerrs, possible_q.shape + numerrs.shape).copy()
    b_tot = np.broadcast_to(numtotal, possible_q.shape + numtotal.shape).copy()
    p = q_to_p(possible_q).astype(np.float64)
    while p.ndim < b_tot.ndim:
        p = np.expand_dims(p, -1)
    b_p = np.broadcast_to(p, b_tot.shape).copy()
    loglike = scipy.stats.binom.logpmf(b_errs + 1, b_tot + 2, b_p)
    assert loglike.shape == prior.shape
    posterior = prior + loglike
    posterior_q = np.argmax(posterior, axis=0)
    return posterior_q - prior_q


def get_delta_qs(meanq, rg_errs, rg_total, q_errs, q_total, pos_errs, pos_total,
                 dinuc_errs, dinuc_total):
    """gatk/applybqsr.py:80-103 -- RG -> Q -> {cycle, dinuc} hierarchy; the dinuc
    table gets one extra zero column so index -1 ("no context") adds 0."""
    rgdq = gatk_delta_q(meanq, rg_errs, rg_total)
    prior1 = np.broadcast_to((meanq + rgdq)[:, np.newaxis], q_total.shape).copy()
    qdq = gatk_delta_q(prior1, q_errs, q_total)
    prior2 = np.broadcast_to((prior1 + qdq)[..., np.newaxis], pos_total.shape).copy()
    posdq = gatk_delta_q(prior2, pos_errs, pos_total)
    prior3 = np.broadcast_to((prior1 + qdq)[..., np.newaxis], dinuc_total.shape).copy()
    ddq = gatk_delta_q(prior3, dinuc_errs, dinuc_total)
    pad = np.zeros((ddq.ndim, 2), dtype=np.int_)
    pad[-1, 1] = 1
    ddq = np.pad(ddq, pad_width=pad, mode='constant', constant_values=0)
    return rgdq.copy(), qdq.copy(), posdq.copy(), ddq.copy()


# --------------------------------------------------------------------------
# read names (compare_reads.py:304-318)
# --------------------------------------------------------------------------
def infer_second(name):
    """compare_reads.py:304-306."""
    return name.split('_')[0][-2:] == '/2'


def infer_rg(name):
    """compare_reads.py:308-318 (IndexError without a second '_' field,
    AssertionError when it does not start with 'RG')."""
    rgstr = name.split('_')[1]
    assert rgstr[0:2] == 'RG'
    return rgstr.split(':')[-1]


# --------------------------------------------------------------------------
# FASTQ text <-> padded planes
# --------------------------------------------------------------------------
def read_fastq(path):
    """Minimal 4-line FASTQ reader with pysam/kseq naming: name = header up to
    the first whitespace (the reference relies on this, recalibrate.py:153)."""
    recs = []
    with open(path, 'rb') as fh:
        lines = fh.read().split(b'\n')
    if lines and lines[-1] == b'':
        lines.pop()
    assert len(lines) % 4 == 0, 'oracle FASTQ reader wants 4-line records'
    for i in range(0, len(lines), 4):
        head = lines[i].decode('ascii')
        assert head[0] == '@'
        name = head[1:].split()[0] if len(head) > 1 else ''
        recs.append((name, lines[i + 1].decode('ascii'), lines[i + 3].decode('ascii')))
    return recs


def pack_records(uncorr, corr, infer_rg_flag, pitch=None):
    """Pair the two record lists the way zip() does (recalibrate.py:57), resolve
    RG ids in first-appearance order (recalibrate.py:59-64) and lay the reads out
    in the padded plane format of kbbq_oracle.c."""
    n = min(len(uncorr), len(corr))
    maxlen = max([len(r[1]) for r in uncorr[:n]] + [1])
    if pitch is None:
        pitch = (maxlen + 15) // 16 * 16
    seq = np.full((n, pitch), ord('N'), dtype=np.uint8)     # layout contract: 'N' past the read ...
    cseq = np.full((n, pitch), ord('N'), dtype=np.uint8)
    qual = np.zeros((n, pitch), dtype=np.uint8)              # ... and 0 in the quality plane
    meta = np.zeros(n, dtype=np.uint32)
    rg_to_int = {}
    for i in range(n):
        name, s, q = uncorr[i]
        cname, cs, _ = corr[i]
        assert cname.startswith(name)                      # recalibrate.py:17
        if len(cs) != len(s):
            raise ValueError('sequence length mismatch')   # numpy broadcast error at :20
        rg = infer_rg(name) if infer_rg_flag else 0
        rgint = rg_to_int.setdefault(rg, len(rg_to_int))
        L = len(s)
        seq[i, :L] = np.frombuffer(s.encode('ascii'), dtype=np.uint8)
        cseq[i, :L] = np.frombuffer(cs.encode('ascii'), dtype=np.uint8)
        qual[i, :L] = np.frombuffer(q.encode('ascii'), dtype=np.uint8)
        meta[i] = L | (rgint << 16) | (int(infer_second(name)) << 31)
    return dict(seq=seq, cseq=cseq, qual=qual, meta=meta, pitch=pitch, n=n,
                S=maxlen if n else 0, R=len(rg_to_int), rg_names=list(rg_to_int))


# --------------------------------------------------------------------------
# C oracle (kbbq_oracle.c) through ctypes
# --------------------------------------------------------------------------
_LIB = None


def build():
    """gcc oracle/kbbq_oracle.c -> oracle/libkbbq_oracle.so (git-ignored)."""
    so = os.path.join(HERE, 'libkbbq_oracle.so')
    src = os.path.join(HERE, 'kbbq_oracle.c')
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['gcc', '-O2', '-std=gnu11', '-fPIC', '-shared',
                               '-ffp-contract=off', '-o', so, src])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.kbbq_oracle_accumulate.restype = ctypes.c_int
        _LIB.kbbq_oracle_apply.restype = ctypes.c_int
        _LIB.kbbq_oracle_synth.restype = None
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _raise(rc, bad):
    if rc == KO_INDEX_ERROR:
        raise IndexError('oracle: index out of bounds at read %d' % bad)
    if rc == KO_TYPE_ERROR:
        raise TypeError('oracle: non-ACGTN base in a dinucleotide at read %d' % bad)
    if rc != 0:
        raise RuntimeError('oracle rc=%d' % rc)


def accumulate(seq, cseq, qual, meta, R, S, minscore=MINSCORE, maxscore=MAXSCORE):
    """recalibrate.py:22-121 on packed planes -> the reference's 9-tuple."""
    n, pitch = seq.shape
    Q = maxscore + 1
    S2 = 2 * S
    ee = np.zeros(R, dtype=np.longdouble)
    rg_e = np.zeros(R, dtype=np.int64); rg_t = np.zeros(R, dtype=np.int64)
    q_e = np.zeros((R, Q), dtype=np.int64); q_t = np.zeros((R, Q), dtype=np.int64)
    p_e = np.zeros((R, Q, S2), dtype=np.int64); p_t = np.zeros((R, Q, S2), dtype=np.int64)
    d_e = np.zeros((R, Q, 16), dtype=np.int64); d_t = np.zeros((R, Q, 16), dtype=np.int64)
    q2p = np.power(10.0, -(np.arange(Q) / 10.0)).astype(np.float64)
    bad = ctypes.c_int64(-1)
    seq = np.ascontiguousarray(seq); cseq = np.ascontiguousarray(cseq)
    qual = np.ascontiguousarray(qual); meta = np.ascontiguousarray(meta)
    rc = lib().kbbq_oracle_accumulate(
        _p(seq), _p(cseq), _p(qual), _p(meta), ctypes.c_int64(n), ctypes.c_int64(pitch),
        ctypes.c_int(R), ctypes.c_int(S2), ctypes.c_int(minscore), ctypes.c_int(maxscore),
        _p(q2p), _p(ee), _p(rg_e), _p(rg_t), _p(q_e), _p(q_t), _p(p_e), _p(p_t),
        _p(d_e), _p(d_t), ctypes.byref(bad))
    _raise(rc, bad.value)
    with np.errstate(divide='ignore', invalid='ignore'):
        meanq = p_to_q(ee / rg_t)                           # recalibrate.py:120
    return meanq, rg_e, rg_t, q_e, q_t, p_e, p_t, d_e, d_t


def apply(seq, qual, meta, meanq, rgdq, qdq, posdq, dinucdq, minscore=MINSCORE):
    """compare_reads.py:320-328 over all reads; returns int32 [n, pitch] raw values."""
    n, pitch = seq.shape
    R, Qt, S2 = posdq.shape
    D = dinucdq.shape[2]
    out = np.zeros((n, pitch), dtype=np.int32)
    a = [np.ascontiguousarray(x, dtype=np.int64) for x in (meanq, rgdq, qdq, posdq, dinucdq)]
    seq = np.ascontiguousarray(seq); qual = np.ascontiguousarray(qual)
    meta = np.ascontiguousarray(meta)
    bad = ctypes.c_int64(-1)
    rc = lib().kbbq_oracle_apply(
        _p(seq), _p(qual), _p(meta), ctypes.c_int64(n), ctypes.c_int64(pitch),
        ctypes.c_int(R), ctypes.c_int(Qt), ctypes.c_int(S2), ctypes.c_int(D),
        ctypes.c_int(minscore), _p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), _p(a[4]),
        _p(out), ctypes.byref(bad))
    _raise(rc, bad.value)
    return out


def synth(first, n, total, seed, len_lo=150, len_hi=150, nrg=1, qlo=0, qhi=41, pitch=None):
    """Synthetic reads, see kbbq_oracle.c."""
    if pitch is None:
        pitch = (len_hi + 15) // 16 * 16
    seq = np.empty((n, pitch), dtype=np.uint8)
    cseq = np.empty((n, pitch), dtype=np.uint8)
    qual = np.empty((n, pitch), dtype=np.uint8)
    meta = np.empty(n, dtype=np.uint32)
    lib().kbbq_oracle_synth(_p(seq), _p(cseq), _p(qual), _p(meta), ctypes.c_int64(first),
                            ctypes.c_int64(n), ctypes.c_int64(total), ctypes.c_int64(pitch),
                            ctypes.c_uint64(seed), ctypes.c_int(len_lo), ctypes.c_int(len_hi),
                            ctypes.c_int(nrg), ctypes.c_int(qlo), ctypes.c_int(qhi),
                            _p(SYNTH_THR))
    return seq, cseq, qual, meta


def synth_names(first, n, nrg=1, with_rg=False):
    """Names matching synth(): r{pair}/{1|2}[_RG:Z:g{pair % nrg}]."""
    out = []
    for i in range(first, first + n):
        nm = 'r%d/%d' % (i >> 1, (i & 1) + 1)
        if with_rg:
            nm += '_RG:Z:g%d' % ((i >> 1) % nrg)
        out.append(nm)
    return out


def write_fastq(path, names, seqplane, qualplane, meta):
    with open(path, 'wb') as fh:
        for i, nm in enumerate(names):
            L = int(meta[i] & 0xFFFF)
            fh.write(b'@' + nm.encode('ascii') + b'\n' + seqplane[i, :L].tobytes() + b'\n+\n'
                     + qualplane[i, :L].tobytes() + b'\n')


# --------------------------------------------------------------------------
# end to end: the text `kbbq recalibrate -f A B [--infer-rg]` prints
# --------------------------------------------------------------------------
def recalibrate_fastq_text(fastq, infer_rg_flag=False):
    """recalibrate.py:123-156 -> the exact stdout text, plus the 9 vectors and the
    4 delta tables for inspection."""
    unc = read_fastq(fastq[0])
    cor = read_fastq(fastq[1])
    b = pack_records(unc, cor, infer_rg_flag)
    vectors = accumulate(b['seq'], b['cseq'], b['qual'], b['meta'], b['R'], b['S'])
    dqs = get_delta_qs(*vectors)
    # pass 2 walks ALL of file A with its own first-appearance RG map (:141-148)
    a = pack_records(unc, unc, infer_rg_flag, pitch=None)
    if a['S'] > b['S'] and a['n']:
        pass  # longer reads in the tail index past the cycle axis -> apply() raises
    newq = apply(a['seq'], a['qual'], a['meta'], vectors[0], *dqs)
    out = []
    for i, (name, s, _) in enumerate(unc):
        L = len(s)
        vals = newq[i, :L].astype(np.int64) + 33
        if np.any(vals < 0) or np.any(vals > 0x10FFFF):
            raise ValueError('recalibrated quality outside the code-point range')
        out.append('@' + name + '\n' + s + '\n+\n' + ''.join(chr(v) for v in vals) + '\n')
    return ''.join(out), vectors, dqs


def sha256(data):
    if isinstance(data, str):
        data = data.encode('utf-8')
    return hashlib.sha256(data).hexdigest()


# --------------------------------------------------------------------------
# slow pure-Python twin of the C tally (small cases only): an independent
# second restatement used to cross-check kbbq_oracle.c in the CPU tests.
# --------------------------------------------------------------------------
_DINUC = {a + b: 4 * i + j for i, a in enumerate('ATGC') for j, b in enumerate('ATGC')}


def py_accumulate(records_uncorr, records_corr, infer_rg_flag=False, minscore=MINSCORE,
                  maxscore=MAXSCORE):
    """recalibrate.py:22-121 with dict/array growth, one base at a time."""
    Q = maxscore + 1
    rg_to_int = {}
    seqlen = 0
    ee = []
    rg_e = []; rg_t = []
    q_e = []; q_t = []
    p_e = []; p_t = []
    d_e = []; d_t = []
    for (name, s, qs), (cname, cs, _) in zip(records_uncorr, records_corr):
        assert cname.startswith(name)
        rg = infer_rg(name) if infer_rg_flag else 0
        if rg not in rg_to_int:
            rg_to_int[rg] = len(rg_to_int)
            ee.append(np.longdouble(0)); rg_e.append(0); rg_t.append(0)
            q_e.append([0] * Q); q_t.append([0] * Q)
            p_e.append([[0] * (2 * seqlen) for _ in range(Q)])
            p_t.append([[0] * (2 * seqlen) for _ in range(Q)])
            d_e.append([[0] * 16 for _ in range(Q)]); d_t.append([[0] * 16 for _ in range(Q)])
        r = rg_to_int[rg]
        L = len(s)
        if L > seqlen:
            grow = 2 * L - 2 * seqlen
            seqlen = L
            for tab in (p_e, p_t):
                for rr in tab:
                    for row in rr:
                        row.extend([0] * grow)          # zeros appended at the END (H1)
        if L < seqlen:
            raise IndexError('boolean index did not match')   # recalibrate.py:97 (H2)
        q = [ord(c) - 33 for c in qs]
        second = infer_second(name)
        for i in range(L):
            cyc = -(i + 1) if second else i
            err = s[i] != cs[i]
            valid = q[i] >= minscore
            if i == 0 or q[i] < minscore or s[i] == 'N' or s[i - 1] == 'N':
                dn = -1
            else:
                dn = _DINUC.get(s[i - 1] + s[i])
                if dn is None:
                    raise TypeError('int() argument must be ... not NoneType')
            if valid:
                ee[r] = ee[r] + np.longdouble(np.power(10.0, -(q[i] / 10.0)))
                rg_t[r] += 1
                q_t[r][q[i]] += 1            # IndexError when q > maxscore
                p_t[r][q[i]][cyc] += 1
                if err:
                    rg_e[r] += 1; q_e[r][q[i]] += 1; p_e[r][q[i]][cyc] += 1
                if dn != -1:
                    d_t[r][q[i]][dn] += 1
                    if err:
                        d_e[r][q[i]][dn] += 1
    R = len(rg_to_int)
    arr = lambda x, shape: np.array(x, dtype=np.int64).reshape(shape)
    ee = np.array(ee, dtype=np.longdouble)
    rg_t_a = arr(rg_t, (R,))
    with np.errstate(divide='ignore', invalid='ignore'):
        meanq = p_to_q(ee / rg_t_a)
    return (meanq, arr(rg_e, (R,)), rg_t_a, arr(q_e, (R, Q)), arr(q_t, (R, Q)),
            arr(p_e, (R, Q, 2 * seqlen)), arr(p_t, (R, Q, 2 * seqlen)),
            arr(d_e, (R, Q, 16)), arr(d_t, (R, Q, 16)))
