"""
Text-format readers for the benchmark path and the BAM-sourced tally (SAM, FASTA, VCF, BED) --
what the reference gets from pysam.AlignmentFile / FastaFile / VariantFile / tabix_iterator
(reference kbbq/benchmark.py:9-30,57-74,145-164; kbbq/gatk/bqsr.py:23-206).  pysam/htslib is not available here:
AlignmentFile reads SAM text, gzip / bgzip-compressed SAM and BAM through the library's own host reader
(csrc/sam_host.cpp, csrc/bam_host.cpp: whole files, no index, no CRAM); BCF and tabix-indexed access are not read.
Objects are duck-typed after pysam so code written against the reference keeps working.
"""

import numpy as np

CIGAR_OPS = 'MIDNSHP=X'


class AlignedRead:
    """The pysam.AlignedSegment attributes the benchmark path uses."""

    def __init__(self, line):
        f = line.rstrip('\r\n').split('\t')
        if len(f) < 11:
            raise ValueError('not a SAM alignment line: %r' % line[:60])
        self.query_name = f[0]
        self.flag = int(f[1])
        self.reference_name = f[2]
        self.reference_start = int(f[3]) - 1
        self.mapping_quality = int(f[4])
        self.cigarstring = f[5]
        self.cigartuples = parse_cigar(f[5])
        self.next_reference_name = f[6]
        self.next_reference_start = int(f[7]) - 1
        self.template_length = int(f[8])
        self.query_sequence = f[9]
        self.query_qualities = [ord(c) - 33 for c in f[10]] if f[10] != '*' else None
        self.tags = {}
        for t in f[11:]:
            k, ty, v = t.split(':', 2)
            self.tags[k] = int(v) if ty == 'i' else v
        self._line = line.rstrip('\r\n')

    @classmethod
    def fromstring(cls, line, header=None):
        return cls(line)

    @property
    def query_length(self):
        return len(self.query_sequence)

    @property
    def reference_end(self):
        return self.reference_start + sum(l for op, l in self.cigartuples if op in (0, 2, 3, 7, 8))

    def _flag_bit(bit):                   # pysam's flag properties: readable and assignable
        def fget(self):
            return bool(self.flag & bit)

        def fset(self, value):
            self.flag = (self.flag | bit) if value else (self.flag & ~bit)
        return property(fget, fset)

    is_paired = _flag_bit(1)
    is_unmapped = _flag_bit(4)
    mate_is_unmapped = _flag_bit(8)
    is_reverse = _flag_bit(16)
    mate_is_reverse = _flag_bit(32)
    is_read1 = _flag_bit(64)
    is_read2 = _flag_bit(128)
    del _flag_bit

    @property
    def tlen(self):                       # pysam's older name of template_length
        return self.template_length

    @tlen.setter
    def tlen(self, v):
        self.template_length = v

    def _soft_clip(self, ops):
        n = 0
        for op, l in ops:
            if op == 4:
                n += l
            elif op != 5:                 # hard clips may sit outside the soft clips
                break
        return n

    @property
    def query_alignment_start(self):
        return self._soft_clip(self.cigartuples)

    @property
    def query_alignment_end(self):
        return self.query_length - self._soft_clip(reversed(self.cigartuples))

    @property
    def query_alignment_length(self):
        return self.query_alignment_end - self.query_alignment_start

    def get_aligned_pairs(self):
        """[(query index | None, reference index | None)] in CIGAR order, soft clips included."""
        out, q, r = [], 0, self.reference_start
        for op, l in self.cigartuples:
            if op in (0, 7, 8):
                out.extend(zip(range(q, q + l), range(r, r + l))); q += l; r += l
            elif op in (1, 4):
                out.extend((i, None) for i in range(q, q + l)); q += l
            elif op in (2, 3):
                out.extend((None, i) for i in range(r, r + l)); r += l
        return out

    def get_tag(self, k):
        return self.tags[k]

    def has_tag(self, k):
        return k in self.tags

    def set_tag(self, k, v):
        self.tags[k] = v

    def __str__(self):
        return self._line


def parse_cigar(text):
    if text == '*':
        return []
    out, num = [], ''
    for ch in text:
        if ch.isdigit():
            num += ch
        else:
            out.append((CIGAR_OPS.index(ch), int(num))); num = ''
    return out


def _read_bytes(path):
    """The whole file as bytes, inflated by the library when it is gzip / bgzip (csrc/sam_host.cpp kbbq_text_open: bgzip blocks side by
    side, gzip members chunk-wise on all host threads -- Python's gzip module reads 0.2-0.3 GB/s on one thread)."""
    import ctypes
    import os
    from . import _native as N
    if not os.path.exists(path):
        raise FileNotFoundError(2, 'No such file or directory', str(path))       # (what open() says)
    lib = N.load()
    handle, data, n = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_size_t(0)
    N.check(lib.kbbq_text_open(str(path).encode(), ctypes.byref(handle), ctypes.byref(data), ctypes.byref(n)))
    try:
        return ctypes.string_at(data, n.value) if n.value else b''
    finally:
        lib.kbbq_text_close(handle)


def _open(path):
    if not str(path).endswith('.gz'):
        return open(path, 'r')
    import io
    return io.TextIOWrapper(io.BytesIO(_read_bytes(path)))


class SamHeader(list):
    """Header lines; as_dict() groups them by record type like pysam's AlignmentHeader."""

    def as_dict(self):
        out = {}
        for line in self:
            fields = line.split('\t')
            if fields[0] == '@CO':
                out.setdefault('CO', []).append('\t'.join(fields[1:]))
                continue
            rec = {x[:2]: x[3:] for x in fields[1:]}
            if fields[0] == '@HD':
                out['HD'] = rec
            else:
                out.setdefault(fields[0][1:], []).append(rec)
        return out


class _SamHandle:
    """Owns the native reader's handle; shared by an AlignmentFile and its SamBatch so that either keeps the
    mapped file alive."""

    def __init__(self, path):
        from . import _native as N
        import ctypes
        self.h = ctypes.c_void_p()
        N.check(N.load().kbbq_sam_open(str(path).encode(), ctypes.byref(self.h)))

    def __del__(self):
        try:
            if self.h:
                from . import _native as N
                N.load().kbbq_sam_close(self.h)
                self.h = None
        except Exception:
            pass


class SamBatch:
    """Every alignment of a SAM file as arrays (the native reader, csrc/sam_host.cpp)."""

    def __init__(self, handle):
        from . import _native as N
        import ctypes
        lib = N.load()
        self._handle = handle                 # keeps the mapping alive for plane() / names()
        native = handle.h
        info = np.zeros(6, dtype=np.int64)
        N.check(lib.kbbq_sam_info(native, N.ptr(info)))
        self.n, ncig, self.maxlen, ncontigs, nrg, _ = (int(x) for x in info)
        n = max(self.n, 1)
        self.flag = np.zeros(n, np.int32); self.contig = np.zeros(n, np.int32)
        self.pos = np.zeros(n, np.int64); self.pnext = np.zeros(n, np.int64); self.tlen = np.zeros(n, np.int64)
        self.qlen = np.zeros(n, np.int32); self.ref_span = np.zeros(n, np.int32); self.clip = np.zeros(n, np.uint32)
        self.cig_off = np.zeros(n, np.uint32); self.cig_n = np.zeros(n, np.uint32); self.rg = np.zeros(n, np.int32)
        hq = np.zeros(n, np.int32)
        N.check(lib.kbbq_sam_fields(native, *[N.ptr(a) for a in (self.flag, self.contig, self.pos, self.pnext, self.tlen,
                                                                self.qlen, self.ref_span, self.clip, self.cig_off,
                                                                self.cig_n, self.rg, hq)]))
        for name in ('flag', 'contig', 'pos', 'pnext', 'tlen', 'qlen', 'ref_span', 'clip', 'cig_off', 'cig_n', 'rg'):
            setattr(self, name, getattr(self, name)[:self.n])
        hq = hq[:self.n]
        self.qual_len = hq & 0xFFFF                       # 0: QUAL is '*'
        self.oq_len = hq >> 16                            # -1: no OQ tag
        self.cigar = np.zeros(max(ncig, 1), np.uint32)
        N.check(lib.kbbq_sam_cigar(native, N.ptr(self.cigar)))
        self.cigar = self.cigar[:ncig]
        self._native = native

        def text(what, i):
            p, ln = ctypes.c_char_p(), ctypes.c_int64(0)
            N.check(lib.kbbq_sam_text(native, what, i, ctypes.byref(p), ctypes.byref(ln)))
            return ctypes.string_at(p, ln.value).decode('latin-1')
        self._text = text
        self.contig_names = [text(3, i) for i in range(ncontigs)]
        self.rg_ids = [text(4, i) for i in range(nrg)]

    def plane(self, which, pitch, first=0, n=None):
        """uint8 [n, pitch]: SEQ (0), QUAL (1) or the OQ tag (2) characters, zero padded."""
        from . import _native as N
        n = self.n - first if n is None else n
        out = np.empty((max(n, 1), pitch), dtype=np.uint8)
        if n == 0:
            out[:] = 0
        N.check(N.load().kbbq_sam_fill(self._native, first, n, pitch, which, N.ptr(out)))
        return out

    def adaptor_trim(self):
        """uint32 [n]: lo | hi << 16, the query positions [lo, hi) past each alignment's adaptor boundary (0: none) -- what
        gatk.bqsr.bamread_adaptor_boundary + _trim_range give read by read (reference bqsr.py:131-206)."""
        from . import _native as N
        out = np.zeros(max(self.n, 1), dtype=np.uint32)
        N.check(N.load().kbbq_sam_adaptor_trim(self._native, N.ptr(out)))
        return out[:self.n]

    def names(self):
        return [self._text(0, i) for i in range(self.n)]

    def line(self, i):
        return self._text(1, i)


class AlignmentFile:
    """A BAM file, a SAM text file or a gzip-compressed SAM file (mode is accepted and ignored): parsed once by the
    native reader (csrc/sam_host.cpp; BAM through csrc/bam_host.cpp: zlib, no htslib) into arrays -- `batch()`, what
    the kernels consume -- and, for code written against pysam, an iterable of AlignedRead objects built on demand
    from the same (for BAM: rendered) SAM lines."""

    def __init__(self, path, mode='r'):
        from . import _native as N
        self._handle = _SamHandle(path)
        self._batch = SamBatch(self._handle)
        info = np.zeros(6, dtype=np.int64)
        N.check(N.load().kbbq_sam_info(self._handle.h, N.ptr(info)))
        self.header = SamHeader(self._batch._text(2, i) for i in range(int(info[5])))
        self._reads = None

    def batch(self):
        return self._batch

    def _objects(self):
        if self._reads is None:
            self._reads = [AlignedRead(self._batch.line(i)) for i in range(self._batch.n)]
        return self._reads

    def __iter__(self):
        return iter(self._objects())

    def __next__(self):
        if not hasattr(self, '_it'):
            self._it = iter(self._objects())
        return next(self._it)

    def __len__(self):
        return self._batch.n

    def close(self):
        """pysam compatibility; the mapping is released when the last user of the handle goes away."""


class FastaFile:
    """A FASTA file read whole (plain or gzip / bgzip-compressed): {name: sequence}; the name is the header line's
    first word.  Records are cut and their line ends removed by bytes operations (a genome is tens of millions of
    lines: a Python loop over them takes half a minute per GB)."""

    def __init__(self, path):
        if str(path).endswith('.gz'):
            data = _read_bytes(path)
        else:
            with open(path, 'rb') as fh:
                data = fh.read()
        self._seqs = {}
        at = 0 if data.startswith(b'>') else data.find(b'\n>') + 1          # text before the first header is ignored
        while 0 <= at < len(data) and data[at:at + 1] == b'>':
            nxt = data.find(b'\n>', at)
            end = len(data) if nxt < 0 else nxt
            eol = data.find(b'\n', at, end)
            head = data[at + 1:end if eol < 0 else eol]
            body = b'' if eol < 0 else data[eol + 1:end]
            words = head.split()
            name = words[0].decode('latin-1') if words else ''
            self._seqs[name] = body.replace(b'\n', b'').replace(b'\r', b'').decode('latin-1')
            at = end + 1
        self.references = list(self._seqs)

    def fetch(self, reference=None):
        return self._seqs[reference]


class VariantRecord:
    def __init__(self, chrom, start, stop):
        self.chrom, self.start, self.stop = chrom, start, stop


def read_vcf(path):
    """Records with pysam's 0-based half-open (start, stop) = (POS - 1, POS - 1 + len(REF))."""
    with _open(path) as fh:
        for line in fh:
            if line.startswith('#') or not line.strip():
                continue
            f = line.split('\t')
            start = int(f[1]) - 1
            yield VariantRecord(f[0], start, start + len(f[3]))


class BedRecord:
    def __init__(self, contig, start, end):
        self.contig, self.start, self.end = contig, start, end


def read_bed(fh):
    """BED intervals from an OPEN text handle (the reference passes argparse's file object)."""
    for line in fh:
        if line.strip() and not line.startswith(('#', 'track', 'browser')):
            f = line.split()
            yield BedRecord(f[0], int(f[1]), int(f[2]))


def chars(text):
    """'ACGT' -> array(['A','C','G','T'], dtype='<U1') without a Python list."""
    return np.frombuffer(text.encode('ascii'), dtype=np.uint8).astype(np.uint32).view('U1')


def codes(arr):
    """Array of 1-character strings (or bytes / str) -> uint8 character codes."""
    if isinstance(arr, (bytes, bytearray)):
        return np.frombuffer(bytes(arr), dtype=np.uint8)
    if isinstance(arr, str):
        return np.frombuffer(arr.encode('ascii'), dtype=np.uint8)
    arr = np.asarray(arr)
    if arr.dtype.kind == 'U':
        return np.ascontiguousarray(arr).view(np.uint32).astype(np.uint8)
    return arr.astype(np.uint8)
