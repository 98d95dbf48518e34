"""
FASTQ text <-> padded SoA planes (host side of the hot path).

pysam is not available on either box, so this module stands in for the two
things the reference uses it for on this path: iterating FASTQ records
(recalibrate.py:56-57,141-142) and the record type itself (``.name``,
``.sequence``, ``.quality``, ``.comment``, ``.get_quality_array()``).  ``name`` is
the header up to the first whitespace, as pysam/kseq define it -- the reference
prints it without the comment (recalibrate.py:153).

`pack_pair` turns the two files into the device layout documented in
include/kbbq_hip.h and performs the host-side checks the reference performs
while it reads: name prefix (recalibrate.py:17), read-group inference in
first-appearance order (recalibrate.py:59-64, compare_reads.py:308-318),
second-in-pair (compare_reads.py:304-306) and the shorter-than-running-max
IndexError (recalibrate.py:89-101).
"""
import numpy as np


class FastxRecord:
    """Duck-type of pysam.FastxRecord (see reference tests/test_compare_reads.py:15-24)."""

    def __init__(self, name=None, sequence=None, quality=None, comment=None):
        self.name = name
        self.sequence = sequence
        self.quality = quality
        self.comment = comment

    def get_quality_array(self, offset=33):
        return [ord(c) - offset for c in self.quality]

    def __str__(self):
        return '@%s\n%s\n+\n%s' % (self.name, self.sequence, self.quality)


class FastqText:
    """A whole 4-line-per-record FASTQ file held as bytes plus line offsets."""

    def __init__(self, path):
        with open(path, 'rb') as fh:
            self.buf = np.frombuffer(fh.read(), dtype=np.uint8)
        nl = np.flatnonzero(self.buf == 10)
        if self.buf.size and (nl.size == 0 or nl[-1] != self.buf.size - 1):
            nl = np.append(nl, self.buf.size)          # last line without '\n'
        if nl.size % 4 != 0:
            raise ValueError('%s: not a 4-line-per-record FASTQ file' % path)
        starts = np.empty(nl.size, dtype=np.int64)
        if nl.size:
            starts[0] = 0
            starts[1:] = nl[:-1] + 1
        ends = nl.astype(np.int64)
        # tolerate CRLF
        cr = (ends > starts) & (self.buf[np.maximum(ends - 1, 0)] == 13)
        ends = ends - cr
        self.n = nl.size // 4
        self.h0, self.h1 = starts[0::4], ends[0::4]
        self.s0, self.s1 = starts[1::4], ends[1::4]
        self.q0, self.q1 = starts[3::4], ends[3::4]
        if self.n and not np.all(self.buf[self.h0] == ord('@')):
            raise ValueError('%s: record header does not start with @' % path)
        if np.any((self.q1 - self.q0) != (self.s1 - self.s0)):
            raise ValueError('%s: sequence and quality lengths differ' % path)

    def names(self):
        b = self.buf.tobytes()
        out = []
        for a, e in zip(self.h0.tolist(), self.h1.tolist()):
            f = b[a + 1:e].split(None, 1)
            out.append(f[0].decode('ascii') if f else '')
        return out

    def lengths(self):
        return (self.s1 - self.s0).astype(np.int64)

    def records(self):
        b = self.buf.tobytes()
        for i in range(self.n):
            head = b[self.h0[i] + 1:self.h1[i]].split(None, 1)
            yield FastxRecord(head[0].decode('ascii') if head else '',
                              b[self.s0[i]:self.s1[i]].decode('ascii'),
                              b[self.q0[i]:self.q1[i]].decode('ascii'),
                              head[1].decode('ascii') if len(head) > 1 else None)

    def plane(self, which, n, pitch):
        """Rows [0, n) of the seq ('s') or qual ('q') lines as a padded [n, pitch] plane
        (padding: 'N' for sequences, 0 for qualities -- include/kbbq_hip.h)."""
        a0 = (self.s0 if which == 's' else self.q0)[:n]
        lens = (self.s1 - self.s0)[:n]
        out = np.full((n, pitch), ord('N') if which == 's' else 0, dtype=np.uint8)
        if n == 0:
            return out
        if np.all(lens == lens[0]) and n > 1 and np.all(np.diff(a0) == a0[1] - a0[0]):
            # uniform records: one strided view, no index arrays
            L = int(lens[0]); stride = int(a0[1] - a0[0])
            view = np.lib.stride_tricks.as_strided(self.buf[int(a0[0]):], shape=(n, L), strides=(stride, 1))
            out[:, :L] = view
            return out
        total = int(lens.sum())
        row = np.repeat(np.arange(n, dtype=np.int64), lens)
        first = np.cumsum(lens) - lens
        col = np.arange(total, dtype=np.int64) - np.repeat(first, lens)
        out[row, col] = self.buf[np.repeat(a0, lens) + col]
        return out


class FastxFile:
    """Context-manager iterator over records, pysam.FastxFile style."""

    def __init__(self, path):
        self._text = FastqText(path)

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def __iter__(self):
        return self._text.records()


def infer_second(name):
    """compare_reads.py:304-306."""
    return name.split(sep='_')[0][-2:] == '/2'


def infer_rg(name):
    """compare_reads.py:308-318."""
    rgstr = name.split(sep='_')[1]
    assert rgstr[0:2] == 'RG'
    return rgstr.split(':')[-1]


def make_meta(names, lens, infer_rg_flag, rg_to_int=None):
    """uint32 sidecar (len | rg << 16 | second << 31); RG ids in first-appearance order.
    Returns (meta, rg_to_int, error) where error = (index, exception) for the first read
    whose read group cannot be inferred (meta is then only valid before that read)."""
    rg_to_int = {} if rg_to_int is None else rg_to_int
    meta = np.zeros(len(names), dtype=np.uint32)
    for i, nm in enumerate(names):
        try:
            rg = infer_rg(nm) if infer_rg_flag else 0
        except (IndexError, AssertionError) as exc:
            return meta, rg_to_int, (i, exc)
        r = rg_to_int.setdefault(rg, len(rg_to_int))
        meta[i] = int(lens[i]) | (r << 16) | (int(infer_second(nm)) << 31)
    return meta, rg_to_int, None


def pitch_for(maxlen):
    return max(16, (int(maxlen) + 15) // 16 * 16)


def pack_pair_py(path_a, path_b, infer_rg_flag):
    """Pure-NumPy twin of pack_pair (kept as the readable statement of the rules and used by
    the tests to check the C++ packer).

    Pass-1 input: reads of file A zipped with file B (recalibrate.py:56-57).

    Host-detectable input errors do not raise here: the reference fails at the FIRST
    offending read, and a device-detected error (TypeError / IndexError) on an earlier
    read must win.  They are returned as ``pending_error = (index, exception, inclusive)``
    -- the caller tallies reads [0, index) (or [0, index] when `inclusive`: the read's own
    device checks precede the host one) and then raises.  Order inside one read, as in
    recalibrate.py:59-119: read-group inference, name prefix, sequence lengths,
    [device: alphabet], shorter-than-running-max, [device: q > 42]."""
    A, B = FastqText(path_a), FastqText(path_b)
    n = min(A.n, B.n)                                   # zip() truncates
    names_a, names_b = A.names(), B.names()
    lens = A.lengths()[:n]
    meta, rg_to_int, rg_err = make_meta(names_a[:n], lens, infer_rg_flag)
    cands = []
    if rg_err is not None:
        cands.append((rg_err[0], 0, rg_err[1]))
    for i in range(n):
        if not names_b[i].startswith(names_a[i]):
            cands.append((i, 1, AssertionError('corrected read %r does not start with %r'
                                               % (names_b[i], names_a[i]))))
            break
    mism = np.flatnonzero((B.s1 - B.s0)[:n] != lens)
    if mism.size:
        cands.append((int(mism[0]), 2, ValueError(
            'operands could not be broadcast together: read %d and its correction differ in length' % mism[0])))
    runmax = np.maximum.accumulate(lens) if n else lens
    short = np.flatnonzero(lens < runmax)
    if short.size:
        cands.append((int(short[0]), 4, IndexError(
            'boolean index did not match indexed array: read %d is shorter than an earlier read '
            '(reference recalibrate.py:89-101)' % short[0])))
    pending = None
    if cands:
        idx, order, exc = min(cands, key=lambda c: (c[0], c[1]))
        pending = (idx, exc, order == 4)
        n = idx + (1 if order == 4 else 0)
        lens = lens[:n]
        meta = meta[:n]
    S = int(lens.max()) if n else 0
    pitch = pitch_for(S)
    R = (int(((meta >> 16) & 0x7FFF).max()) + 1) if n else 0
    return dict(seq=A.plane('s', n, pitch), cseq=B.plane('s', n, pitch), qual=A.plane('q', n, pitch),
                meta=meta, n=n, pitch=pitch, S=S, R=R, rg_to_int=rg_to_int,
                names=names_a, text=A, pending_error=pending)


def pack_single_py(text, infer_rg_flag):
    """Pass-2 input: every read of file A with its own first-appearance RG map
    (recalibrate.py:141-148)."""
    names = text.names()
    lens = text.lengths()
    meta, rg_to_int, err = make_meta(names, lens, infer_rg_flag)
    if err is not None:
        raise err[1]
    S = int(lens.max()) if text.n else 0
    pitch = pitch_for(S)
    return dict(seq=text.plane('s', text.n, pitch), qual=text.plane('q', text.n, pitch), meta=meta,
                n=text.n, pitch=pitch, S=S, R=len(rg_to_int), names=names)


def format_fastq(names, seq_plane, qual_plane, lens):
    """recalibrate.py:153-156: '@'+name, sequence, '+', qualities -- one str."""
    n = len(names)
    if n == 0:
        return ''
    parts = []
    sb = seq_plane.tobytes()
    qb = qual_plane.tobytes()
    pitch = seq_plane.shape[1]
    for i in range(n):
        L = int(lens[i])
        o = i * pitch
        parts.append('@' + names[i] + '\n' + sb[o:o + L].decode('latin-1') + '\n+\n'
                     + qb[o:o + L].decode('latin-1') + '\n')
    return ''.join(parts)


# ---------------------------------------------------------------------------
# C++ packer / writer (csrc/fastq_host.cpp through the C ABI): multi-threaded, mmap-based.
# Same rules and the same result dictionaries as the NumPy twins above.
# ---------------------------------------------------------------------------
import ctypes as _ct

from . import _native as _N


class NativeFastq:
    """A FASTQ file opened by libkbbq_hip's host reader (kbbq_fastq_*)."""

    def __init__(self, path, _handle=None):
        self._h = _ct.c_void_p() if _handle is None else _handle
        if _handle is None:
            _N.check(_N.load().kbbq_fastq_open(str(path).encode(), _ct.byref(self._h)))
        self.n = int(_N.load().kbbq_fastq_count(self._h))
        # a reader of one byte range of the file (open_range) holds records [first, first + n) of `total`; every method
        # below takes FILE-WIDE record numbers
        self.first, self.total = 0, self.n

    @classmethod
    def open_range(cls, path, byte_lo, byte_hi, first, total, rg_names=()):
        """Only the records in the byte range [byte_lo, byte_hi) of an uncompressed file -- records first, first + 1,
        ... of its `total` -- with the file-wide read-group names of the rank that scanned all of it."""
        h = _ct.c_void_p()
        _N.check(_N.load().kbbq_fastq_open_range(str(path).encode(), int(byte_lo), int(byte_hi), _ct.byref(h)))
        f = cls(None, _handle=h)
        f.first, f.total = int(first), int(total)
        blob = b''.join(str(x).encode('ascii') + b'\0' for x in rg_names)
        _N.check(_N.load().kbbq_fastq_set_rg_names(h, blob, len(rg_names)))
        return f

    def record_offset(self, i):
        """Byte offset of record i's '@' (i = first + n: the end of the indexed range)."""
        off = int(_N.load().kbbq_fastq_record_offset(self._h, i - self.first))
        if off < 0:
            _N.check(_N.KBBQ_E_ARG)
        return off

    def is_plain(self):
        return bool(_N.load().kbbq_fastq_is_plain(self._h))

    _h = None                                # (class default: what a closed reader's handle reads as)

    def close(self):
        h = self.__dict__.pop('_h', None)    # one atomic step: a close on a helper thread (close_later) and another here cannot both get it
        if h:
            _N.load().kbbq_fastq_close(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def name(self, i):
        p, ln = _ct.c_void_p(), _ct.c_int(0)
        _N.check(_N.load().kbbq_fastq_name(self._h, i - self.first, _ct.byref(p), _ct.byref(ln)))
        return _ct.string_at(p, ln.value).decode('ascii')

    def names(self):
        return [self.name(self.first + i) for i in range(self.n)]

    def rg_names(self):
        lib = _N.load()
        return [lib.kbbq_fastq_rg_name(self._h, i).decode('ascii') for i in range(lib.kbbq_fastq_rg_count(self._h))]

    def scan_next(self, other, infer_rg_flag, rg_names, prior_longest):
        """scan() of one segment of a sequentially read input: `rg_names` are the read groups of the segments before it
        (new ones are appended: rg_names() afterwards), prior_longest their longest read (kbbq_fastq_scan_next)."""
        blob = b''.join(str(x).encode('ascii') + b'\0' for x in rg_names)
        _N.check(_N.load().kbbq_fastq_set_rg_names(self._h, blob, len(rg_names)))
        info = np.zeros(5, dtype=np.int64)
        _N.check(_N.load().kbbq_fastq_scan_next(self._h, other._h if other is not None else None, 1 if infer_rg_flag else 0,
                                                int(prior_longest), _N.ptr(info)))
        return [int(x) for x in info]

    def scan(self, other, infer_rg_flag):
        info = np.zeros(5, dtype=np.int64)
        _N.check(_N.load().kbbq_fastq_scan(self._h, other._h if other is not None else None,
                                           1 if infer_rg_flag else 0, _N.ptr(info)))
        return [int(x) for x in info]

    def lengths(self, first=None, n=None):
        first = self.first if first is None else first
        n = self.first + self.n - first if n is None else n
        out = np.zeros(max(n, 1), dtype=np.uint32)
        _N.check(_N.load().kbbq_fastq_lengths(self._h, first - self.first, n, _N.ptr(out)))
        return out[:n]

    def length_bands(self, first=None, n=None, max_bands=16):
        """length_bands (below) of reads [first, first + n), computed by the reader (lo / hi relative to first)."""
        first = self.first if first is None else first
        n = self.first + self.n - first if n is None else n
        classes = np.asarray(BAND_CLASSES, dtype=np.uint32)
        out = np.zeros((max_bands, 4), dtype=np.int64)
        runs = _N.load().kbbq_fastq_length_runs(self._h, first - self.first, n, _N.ptr(classes), len(classes), max_bands, _N.ptr(out))
        if runs < 0:
            _N.check(runs)
        return [tuple(int(x) for x in row) for row in out[:runs]]

    def fill(self, other, infer_rg_flag, n, pitch, first=0):
        """Planes and sidecar of reads [first, first + n) (rows 0..n-1)."""
        seq = np.empty((n, pitch), dtype=np.uint8)
        qual = np.empty((n, pitch), dtype=np.uint8)
        cseq = np.empty((n, pitch), dtype=np.uint8) if other is not None else None
        meta = np.empty(n, dtype=np.uint32)
        self.fill_into(other, infer_rg_flag, n, pitch, first, seq, cseq, qual, meta)
        return seq, cseq, qual, meta

    def fill_into(self, other, infer_rg_flag, n, pitch, first, seq, cseq, qual, meta):
        """The same into caller-provided C-contiguous arrays ([n, pitch] uint8 planes, [n] uint32 sidecar), e.g. views
        of page-locked staging buffers."""
        for a, shape in ((seq, (n, pitch)), (qual, (n, pitch)), (cseq, (n, pitch)), (meta, (n,))):
            if a is not None and (tuple(a.shape) != shape or not a.flags['C_CONTIGUOUS']):
                raise ValueError('fill_into: array of shape %s, expected C-contiguous %s' % (a.shape, shape))
        if other is not None and other.first != self.first:
            raise ValueError('the two readers of a pair must start at the same record')
        _N.check(_N.load().kbbq_fastq_fill_range(self._h, other._h if other is not None else None,
                                                 1 if infer_rg_flag else 0, first - self.first, n, pitch, _N.ptr(seq),
                                                 _N.ptr(cseq), _N.ptr(qual), _N.ptr(meta)))

    def meta(self, infer_rg_flag, first=None, n=None):
        """(sidecar words uint32 [n], statistics) of reads [first, first + n): kbbq_fastq_meta -- the host twin of
        _device.meta_stats, computed from the text before anything is uploaded."""
        first = self.first if first is None else first
        n = self.first + self.n - first if n is None else n
        meta = np.empty(max(n, 1), dtype=np.uint32)
        st = np.zeros(8, dtype=np.int32)
        _N.check(_N.load().kbbq_fastq_meta(self._h, 1 if infer_rg_flag else 0, first - self.first, n, _N.ptr(meta), _N.ptr(st)))
        return meta[:n], {'shortest': int(st[0]) if st[0] != 0x7FFFFFFF else 0, 'longest': int(st[1]), 'max_rg': int(st[2]),
                          'pair_violations': int(st[3]), 'empty': int(st[4]), 'twin_violations': int(st[5])}

    def fill_rows(self, other, first, n, meta, flags, S2, pitch, perm, row_lo, nrows, seq, cseq, qual, dmeta):
        """Destination rows [row_lo, row_lo + nrows) of reads [first, first + n) in the layout `flags` (the _native.ROWS_*
        bits) straight from the text (kbbq_fastq_fill_rows).  Returns True when a letter outside ACGTN met the nibble
        packing (the band then needs character planes)."""
        if other is not None and other.first != self.first:
            raise ValueError('the two readers of a pair must start at the same record')
        foreign = _ct.c_int(0)
        _N.check(_N.load().kbbq_fastq_fill_rows(self._h, other._h if other is not None else None, first - self.first, n,
                                                 _N.ptr(meta), int(flags), int(S2), int(pitch), _N.ptr(perm), int(row_lo), int(nrows),
                                                 _N.ptr(seq), _N.ptr(cseq), _N.ptr(qual), _N.ptr(dmeta), _ct.byref(foreign)))
        return bool(foreign.value)

    def format_rows_array(self, first, n, newqual, flags, S2, out=None):
        """format_array with the new qualities still in the rows K2 wrote (mate-pair rows when flags has ROWS_PAIRS:
        `first` is the first mate of newqual's row 0)."""
        newqual = np.ascontiguousarray(newqual)
        pitch = newqual.shape[1]
        lib = _N.load()
        local = first - self.first
        need = int(-lib.kbbq_fastq_format_rows(self._h, local, n, int(flags), int(S2), pitch, _N.ptr(newqual), None, 0))
        if need == 0 and n:
            _N.check(_N.KBBQ_E_ARG)
        buf = np.empty(max(need, 1), dtype=np.uint8) if out is None else out(max(need, 1))
        got = lib.kbbq_fastq_format_rows(self._h, local, n, int(flags), int(S2), pitch, _N.ptr(newqual), _N.ptr(buf), need)
        if got != need:
            _N.check(_N.KBBQ_E_ARG)
        return buf[:need]

    def format_array(self, first, n, newqual, out=None):
        """FASTQ text of reads [first, first+n) with qualities from rows of `newqual`, as a uint8 array
        (written once by the C++ writer: no zero fill, no copy).  out: callable nbytes -> uint8 array of at least
        that size to render into (a re-used buffer) instead of a fresh one."""
        newqual = np.ascontiguousarray(newqual)
        pitch = newqual.shape[1]
        lib = _N.load()
        local = first - self.first
        need = int(-lib.kbbq_fastq_format(self._h, local, n, pitch, _N.ptr(newqual), None, 0))
        buf = np.empty(max(need, 1), dtype=np.uint8) if out is None else out(max(need, 1))
        got = lib.kbbq_fastq_format(self._h, local, n, pitch, _N.ptr(newqual), _N.ptr(buf), need)
        assert got == need
        return buf[:need]

    def format(self, first, n, newqual):
        """The same as bytes."""
        return self.format_array(first, n, newqual).tobytes()


class FastqStream:
    """A FASTQ input read sequentially, segment by segment (kbbq_fastq_stream_*: csrc/fastq_stream.cpp) -- a regular file, a
    named pipe / process substitution, or '-' for standard input.  next() returns (NativeFastq over the segment's own
    memory or None at the end, at_end); the segment's `first` / `total` number its records file-wide as far as known."""

    def __init__(self, path):
        self._h = _ct.c_void_p()
        self.path = str(path)
        _N.check(_N.load().kbbq_fastq_stream_open(self.path.encode(), _ct.byref(self._h)))
        self.regular = bool(_N.load().kbbq_fastq_stream_is_regular(self._h))
        self.records = 0                     # handed out so far

    def tee(self, fd):
        _N.check(_N.load().kbbq_fastq_stream_tee(self._h, int(fd)))

    def prefetch(self, nbytes):
        """Read up to nbytes ahead of the next next() (on another thread, while the other file's segment is being read)."""
        _N.check(_N.load().kbbq_fastq_stream_prefetch(self._h, int(nbytes)))

    def next(self, max_bytes, records=0):
        seg, end = _ct.c_void_p(), _ct.c_int(0)
        _N.check(_N.load().kbbq_fastq_stream_next(self._h, int(max_bytes), int(records), _ct.byref(seg), _ct.byref(end)))
        if not seg.value:
            return None, True
        f = NativeFastq(None, _handle=seg)
        f.first, f.total = self.records, self.records + f.n
        self.records += f.n
        return f, bool(end.value)

    _h = None

    def close(self):
        h = self.__dict__.pop('_h', None)
        if h:
            _N.load().kbbq_fastq_stream_close(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


GZ_STREAM_BYTES = 1 << 30               # a compressed file of at least this size is inflated segment by segment (KBBQ_GZ_STREAM_BYTES)


def is_sequential_input(path):
    """True for an input that is read sequentially instead of being mapped and indexed as a whole: standard input, a pipe, a
    character device -- and a gzip / bgzip-compressed regular file of 1 GB or more (KBBQ_GZ_STREAM_BYTES), which the mapped
    reader would have to inflate into memory as a whole (a 100 GB .fq.gz is 300 GB of text; pysam streams it, so does
    FastqStream).  Smaller compressed files keep the mapped reader: bgzip blocks are inflated in parallel there."""
    import os
    import stat
    if str(path) == '-':
        return True
    try:
        st = os.stat(str(path))
    except OSError:
        return False
    if not stat.S_ISREG(st.st_mode):
        return True
    limit = os.environ.get('KBBQ_GZ_STREAM_BYTES')
    limit = GZ_STREAM_BYTES if not limit else int(float(limit))
    return st.st_size >= limit and st.st_size >= 2 and _is_gzip(str(path))


_SCAN_ERRORS = {
    1: lambda i: IndexError('list index out of range'),                       # name.split('_')[1]
    2: lambda i: AssertionError('read %d: second name field does not start with RG' % i),
    3: lambda i: AssertionError('read %d: corrected read name does not start with the read name' % i),
    4: lambda i: ValueError('operands could not be broadcast together: read %d and its correction '
                            'differ in length' % i),
    5: lambda i: IndexError('boolean index did not match indexed array: read %d is shorter than an '
                            'earlier read (reference recalibrate.py:89-101)' % i),
}


# Row widths offered to the length bands of a mixed-length input: a read goes into the narrowest class that holds
# it, so the bytes moved per read follow its own length instead of the longest read's (a 50-base read in a
# 304-byte row costs 6x).  Few classes: every band is one kernel launch.
BAND_CLASSES = (32, 48, 64, 96, 128, 160, 208, 256, 320, 400, 512, 768, 1024, 2048, 4096, 16384, 65536)


def length_bands(lens, max_bands=16):
    """[(lo, hi, longest, shortest)] -- maximal runs of reads whose lengths fall into one BAND_CLASSES class
    (shortest: of the non-empty reads, 0 when there is none).  Valid inputs
    have non-decreasing lengths, i.e. at most one run per class; anything with more than max_bands
    runs is kept as one band."""
    lens = np.asarray(lens, dtype=np.int64)
    n = len(lens)
    if n == 0:
        return []
    cls = np.searchsorted(np.asarray(BAND_CLASSES), np.maximum(lens, 1), side='left')
    cuts = np.flatnonzero(np.diff(cls)) + 1
    if len(cuts) + 1 > max_bands:
        cuts = cuts[:0]
    edges = [0] + cuts.tolist() + [n]
    out = []
    for lo, hi in zip(edges[:-1], edges[1:]):
        part = lens[lo:hi]
        some = part[part > 0]
        out.append((lo, hi, int(part.max()), int(some.min()) if len(some) else 0))
    return out


def _fill_bands(A, B, infer_rg_flag, lo, hi, to_device=False, R=1, S=None):
    """Reads [lo, hi) packed band by band: [dict(first, n, S, Smin, pitch, seq, cseq, qual, meta)], `first` counted from
    lo.  to_device: instead of host planes every band carries `laid`, the band on the device in the layout the kernels
    run fastest on, written by the packer itself slab by slab through page-locked staging (_device.laid_from_reader;
    R = read groups, S = the input's longest read: mate-pair rows need count tables of exactly 2 x the band's length) --
    or, when no layout applies, `batch`: one character row per read (_device.ReadBatch.from_reader).  `source` lets
    band_rows() make those rows later for a band that has `laid` only (whatever a layout's kernels refuse is redone on
    them: they carry the reference's exact error semantics)."""
    from ._trace import stage
    out = []
    with stage('bands'):
        bands = A.length_bands(lo, hi - lo)
    for b_lo, b_hi, longest, shortest in bands:
        pitch = pitch_for(longest)
        band = dict(first=b_lo, n=b_hi - b_lo, S=longest, Smin=shortest, pitch=pitch)
        if to_device:
            from . import _device as dev
            band['source'] = (A, B, infer_rg_flag, lo + b_lo)
            band['batch'] = None
            band['laid'] = dev.laid_from_reader(A, B, infer_rg_flag, lo + b_lo, b_hi - b_lo, pitch, max(R, 1),
                                                packed=longest <= dev.PACKED_READS, pairs=None if S in (None, longest) else False,
                                                with_out=True)
            if band['laid'] is None:
                band_rows(band)
        else:
            seq, cseq, qual, meta = A.fill(B, infer_rg_flag, b_hi - b_lo, pitch, first=lo + b_lo)
            band.update(seq=seq, cseq=cseq, qual=qual, meta=meta)
        out.append(band)
    return out


def band_rows(band):
    """The band as one character row per read on the device (band['batch']), filled from the text on first use."""
    if band.get('batch') is None:
        from . import _device as dev
        A, B, infer_rg_flag, first = band['source']
        if B is not None and (getattr(B, '_closing', False) or B._h is None):
            B = None                                              # the corrected file has been closed: pass 2 does not need it
        band['batch'] = dev.ReadBatch.from_reader(A, B, infer_rg_flag, first, band['n'], band['pitch'], keep_pinned=band.get('keep_pinned', False))
    return band['batch']


def _too_large(n, S, budget):
    """Would n reads of up to S bases, resident on the device (three input planes, the output plane, sidecars), need more
    than `budget` bytes?  (kbbq/_stream.py resident_bytes)"""
    return int(n) * (4 * pitch_for(S) + 4) > int(budget)


def _shard(n, shard):
    """Records [lo, hi) of this rank among n (pairs stay together), or everything."""
    if shard is None:
        return 0, n
    from .parallel import shard_range
    return shard_range(n, shard[0], shard[1])


def close_later(reader):
    """Close a NativeFastq on a helper thread: unmapping GBs of file costs tens of milliseconds, which need not sit on
    the path to the first kernel (the C call runs without the interpreter lock)."""
    import threading
    reader._closing = True                   # from here on nobody may start new work on it (band_rows)
    if LEAVE_OPEN:
        _left_open.append(reader)
        return
    threading.Thread(target=reader.close, daemon=True).start()


# The command line as a single process ends with os._exit right after its last byte (main._leave): unmapping GBs of input -- on a
# helper thread or not -- only makes everything else that needs the address space's lock wait (the interpreter freeing its arrays
# on the way out: 80 ms); the readers are parked here and go with the process.
LEAVE_OPEN = False
_left_open = []


def retire(reader):
    """No new work may start on the reader (band_rows treats it as closed); it stays mapped until somebody closes it."""
    reader._closing = True
    return reader


class PairScan:
    """Both files of a pair being opened (mapped, line-indexed) and scanned by the C++ reader on its own threads --
    started by the constructor, which returns at once; result() waits.  The work needs no interpreter lock, so a
    caller can import torch and set the device up meanwhile (recalibrate._pack_and_tally)."""

    def __init__(self, path_a, path_b, infer_rg_flag):
        self._job = _ct.c_void_p()
        _N.check(_N.load().kbbq_fastq_pair_begin(str(path_a).encode(), None if path_b is None else str(path_b).encode(),
                                                 1 if infer_rg_flag else 0, _ct.byref(self._job)))
        self._has_b = path_b is not None

    def result(self):
        """(A, B or None, [total, S, R, kind, idx]); raises what opening file A, then file B, would have."""
        job, self._job = self._job, None
        if job is None:
            raise RuntimeError('PairScan.result() may be called once')
        a, b, info = _ct.c_void_p(), _ct.c_void_p(), np.zeros(5, dtype=np.int64)
        _N.check(_N.load().kbbq_fastq_pair_wait(job, _ct.byref(a), _ct.byref(b), _N.ptr(info)))
        return NativeFastq(None, _handle=a), (NativeFastq(None, _handle=b) if self._has_b else None), [int(x) for x in info]

    def discard(self):
        """Join the job and drop its readers (the caller opened the files another way)."""
        if getattr(self, '_job', None) is not None:
            try:
                self.result()
            except Exception:
                pass

    def __del__(self):
        try:
            if getattr(self, '_job', None) is not None:
                self.result()                # joins the threads; the readers close with their wrappers
        except Exception:
            pass


def _shard_plan(A, B, scanned, world):
    """What rank 0 tells the others after it has indexed and scanned the pair: the scan's result, the file-wide
    read-group names and, per rank, its records [lo, hi) with the byte ranges that hold them in either file."""
    total, S, R, kind, idx = scanned
    ranges = []
    for r in range(world):
        lo, hi = _shard(total, (r, world))
        ranges.append((lo, hi, A.record_offset(lo), A.record_offset(hi), B.record_offset(lo), B.record_offset(hi)))
    return dict(scanned=[total, S, R, kind, idx], rgs=A.rg_names(), n_text=A.total, ranges=ranges,
                plain=A.is_plain() and B.is_plain())


def _is_gzip(path):
    with open(path, 'rb') as fh:
        return fh.read(2) == b'\x1f\x8b'


def local_ranges(path_a, path_b, infer_rg_flag, rank, world, gather):
    """The multi-rank opening of a pair in which NOBODY reads a whole file: every rank cuts its own
    byte range out of either file (cut r = the first record start at or after size * r / world that is not a
    second-in-pair record: kbbq_fastq_sync_offset_ex, computed identically by both ranks that share the cut), indexes
    and scans only that, and ONE all_gather_object carries what is file-wide: record counts (-> every rank's first
    record number), read-group names in first-appearance order (merged in rank order), the longest read, the first
    read's length (a first read shorter than the longest read of the ranks before it is the reference's IndexError,
    recalibrate.py:89-101) and the first offender each rank's own scan found.
    Returns (A, B, [usable, S, R, kind, idx]) with A / B range readers numbered file-wide, or None when the two files
    do not cut into the same records (different record counts per range: the corrected file is shorter, or its lines
    have other lengths; compressed input) -- the caller then falls back to the plan of a rank that scans everything.
    `gather`: obj -> the list of every rank's obj (parallel.all_gather_object)."""
    import os
    lib = _N.load()
    plain = not _is_gzip(path_a) and not _is_gzip(path_b)
    A = B = None
    mine = dict(plain=plain, nA=0, nB=0, scan=None, rgs=[], len0=0, error=None)
    if plain:
        try:
            def cuts(path):
                size = os.path.getsize(path)
                lo, hi = (int(lib.kbbq_fastq_sync_offset_ex(str(path).encode(), size * r // world, 1)) if 0 < r < world
                          else (0 if r == 0 else size) for r in (rank, rank + 1))
                if lo < 0 or hi < 0:
                    _N.check(_N.KBBQ_E_ARG)
                return lo, max(hi, lo)
            a_lo, a_hi = cuts(path_a)
            A = NativeFastq.open_range(path_a, a_lo, a_hi, 0, 0)
            b_lo, b_hi = cuts(path_b)
            B = NativeFastq.open_range(path_b, b_lo, b_hi, 0, 0)
            mine.update(nA=A.n, nB=B.n)
            if A.n == B.n:
                mine['scan'] = A.scan(B, infer_rg_flag)
                mine['rgs'] = A.rg_names()
                mine['len0'] = int(A.lengths(0, 1)[0]) if A.n else 0
        except Exception as e:               # noqa: BLE001 -- every rank must reach the gather
            mine['error'] = (type(e).__name__, str(e))
    everyone = gather(mine)
    for o in everyone:                                           # a rank that could not read its range stops them all
        if o['error'] is not None:
            import builtins
            raise getattr(builtins, o['error'][0], RuntimeError)(o['error'][1])
    if not all(o['plain'] and o['nA'] == o['nB'] and o['scan'] is not None for o in everyone):
        for f in (A, B):
            if f is not None:
                f.close()
        return None
    firsts = np.concatenate([[0], np.cumsum([o['nA'] for o in everyone])]).astype(np.int64)
    total = int(firsts[-1])
    # the first offender, file-wide: every rank's own (made file-wide), and a first read shorter than what came before
    err = None                                                   # (index, kind)
    longest_before = 0
    S = 0
    order = {}
    for r, o in enumerate(everyone):
        usable_r, S_r, R_r, kind_r, idx_r = o['scan']
        cand = []
        if kind_r:
            cand.append((int(firsts[r]) + idx_r, kind_r))
        if o['nA'] and o['len0'] < longest_before:
            cand.append((int(firsts[r]), 5))
        for nm in o['rgs']:
            order.setdefault(nm, len(order))
        if cand:
            err = min(cand)
            S = max(S, S_r if err == (int(firsts[r]) + idx_r, kind_r) and kind_r else o['len0'])
            break
        S = max(S, S_r)
        longest_before = max(longest_before, S_r)
    usable = total if err is None else err[0] + (1 if err[1] == 5 else 0)
    rgs = list(order) if infer_rg_flag else (['0'] if total else [])
    R = (len(rgs) if infer_rg_flag else 1) if usable > 0 else 0
    blob = b''.join(str(x).encode('ascii') + b'\0' for x in rgs)
    for f in (A, B):
        f.first, f.total = int(firsts[rank]), total
        _N.check(lib.kbbq_fastq_set_rg_names(f._h, blob, len(rgs)))
    return A, B, [usable, S, R, err[1] if err else 0, err[0] if err else -1]


def pack_pair(path_a, path_b, infer_rg_flag, shard=None, bands=False, scan=None, to_device=False, exchange=None, gather=None, budget=None):
    """Pass-1 input (recalibrate.py:56-57) through the C++ packer; same dictionary as pack_pair_py
    except that `text` is the NativeFastq of file A and `names` is filled lazily by callers.
    shard = (rank, world): the rank packs only its own records [first, first + n); `total` is the global number of
    usable reads.  Read-group ids, the longest read and the first host-detectable error are file-wide facts:
    without `exchange` every rank opens and scans the whole pair itself; with it (a function that returns rank 0's
    object on every rank: parallel.broadcast_object) only rank 0 does, and the others index just the byte ranges of
    their shard (uncompressed files; compressed ones fall back to everybody reading everything).
    bands=True: instead of one set of planes at the widest pitch, `bands` holds the reads packed by length band
    (length_bands), each at its own pitch.  scan: a PairScan of the same arguments started earlier.  to_device (with
    bands): the bands are filled straight onto the device (_fill_bands) -- unless `budget` (bytes of device memory) is given
    and the shard's resident form would need more: then nothing is filled, `bands` is empty and `streamed` names the readers
    and the records for kbbq/_stream.py, which walks them slab by slab in either pass."""
    from ._trace import stage
    rank, world = shard if shard is not None else (0, 1)
    planned = exchange is not None and world > 1
    A = B = scanned = None
    failure = None
    own_range = None                         # (first, n): this rank's records when the ranks cut their own byte ranges
    if gather is not None and world > 1:
        with stage('open+index+scan (own byte range)'):
            got = local_ranges(path_a, path_b, infer_rg_flag, rank, world, gather)
        if got is not None:
            A, B, scanned = got
            own_range = (A.first, A.n)
            planned = False
            if scan is not None:
                scan.discard()
    if own_range is not None:
        pass
    elif not planned or rank == 0:
        try:
            with stage('open+index+scan (wait)'):
                A, B, scanned = (scan or PairScan(path_a, path_b, infer_rg_flag)).result()
        except Exception as e:               # noqa: BLE001 -- with a plan to hand out, the other ranks must hear of it
            if not planned:
                raise
            failure = e
    if planned:
        if rank == 0:
            plan = exchange(dict(error=(type(failure).__name__, str(failure))) if failure else _shard_plan(A, B, scanned, world))
        else:
            plan = exchange(None)
        if failure is not None:
            raise failure
        if 'error' in plan:                                      # rank 0 could not read the files: everybody stops
            import builtins
            raise getattr(builtins, plan['error'][0], RuntimeError)(plan['error'][1])
        if rank != 0 and plan['plain']:
            scanned = plan['scanned']
            lo, hi, a_lo, a_hi, b_lo, b_hi = plan['ranges'][rank]
            with stage('open+index (shard)'):
                A = NativeFastq.open_range(path_a, a_lo, a_hi, lo, plan['n_text'], plan['rgs'])
                B = NativeFastq.open_range(path_b, b_lo, b_hi, lo, plan['n_text'], plan['rgs'])
            if (A.n, B.n) != (hi - lo, hi - lo):
                raise ValueError('the shard of rank %d holds %d / %d records, %d expected: the files changed while '
                                 'being read' % (rank, A.n, B.n, hi - lo))
        elif rank != 0:                                          # compressed input: every rank reads all of it
            with stage('open+index+scan (wait)'):
                A, B, scanned = PairScan(path_a, path_b, infer_rg_flag).result()
    total, S, R, kind, idx = scanned
    pending = (idx, _SCAN_ERRORS[kind](idx), kind == 5) if kind else None
    pitch = pitch_for(S)
    if own_range is not None:                # this rank's own records, cut at the first offender
        lo = min(own_range[0], total)
        hi = min(own_range[0] + own_range[1], total)
        if hi <= lo:                         # everything this rank holds lies behind the first offender
            lo = hi = own_range[0]
    else:
        lo, hi = _shard(total, shard)
    rgs = A.rg_names()
    common = dict(n=hi - lo, first=lo, total=total, pitch=pitch, S=S, R=R,
                  rg_to_int={(nm if infer_rg_flag else 0): i for i, nm in enumerate(rgs)},
                  text=A, pending_error=pending)
    keep = False
    try:
        with stage('fill'):
            if bands and to_device and budget is not None and hi > lo and _too_large(hi - lo, S, budget):
                keep = True
                return dict(common, other=B, bands=[], streamed=dict(A=A, B=B, infer_rg=infer_rg_flag, lo=lo, hi=hi, budget=int(budget)))
            if bands:
                filled = _fill_bands(A, B, infer_rg_flag, lo, hi, to_device, R, S)
                keep = to_device             # only once the bands exist (a fill that raises must not leak B's mapping: ADVICE r3); a
                return dict(common, other=B if to_device else None, bands=filled)    # band the kernels refuse in its layout is re-filled
                                                                                     # as character rows: the caller closes B after the tally
            seq, cseq, qual, meta = A.fill(B, infer_rg_flag, hi - lo, pitch, first=lo)
        return dict(common, seq=seq, cseq=cseq, qual=qual, meta=meta)
    finally:
        if not keep:
            close_later(B)                   # file B is not needed after the fill: unmap it off the critical path


def pack_single(text, infer_rg_flag, shard=None, bands=False, to_device=False, budget=None):
    """Pass-2 input: every read of file A with its own first-appearance RG map
    (recalibrate.py:141-148); shard = (rank, world) as in pack_pair."""
    total, S, R, kind, idx = text.scan(None, infer_rg_flag)
    if kind:
        raise _SCAN_ERRORS[kind](idx)
    pitch = pitch_for(S)
    lo, hi = _shard(total, shard)
    rgs = text.rg_names()
    common = dict(n=hi - lo, first=lo, total=total, pitch=pitch, S=S, R=R,
                  rg_to_int={(nm if infer_rg_flag else 0): i for i, nm in enumerate(rgs)})
    if bands and to_device and budget is not None and hi > lo and _too_large(hi - lo, S, budget):
        return dict(common, bands=[], streamed=dict(A=text, B=None, infer_rg=infer_rg_flag, lo=lo, hi=hi, budget=int(budget)))
    if bands:
        return dict(common, bands=_fill_bands(text, None, infer_rg_flag, lo, hi, to_device, R, S))
    seq, _, qual, meta = text.fill(None, infer_rg_flag, hi - lo, pitch, first=lo)
    return dict(common, seq=seq, qual=qual, meta=meta)
