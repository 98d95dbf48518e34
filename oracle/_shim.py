"""
_shim.py -- lets the UNMODIFIED reference (/root/reference/kbbq) import in the
build container.  TEST INFRASTRUCTURE ONLY; used by oracle/gen_golden.py and
nothing else.  It contains no reference code.

Why it is needed: the reference imports pysam, khmer and
seaborn, none of which are installed, and uses NumPy aliases that NumPy 2.x
removed (np.int, np.bool, np.float, np.object, np.unicode, np.NINF) and one pandas method
that pandas 2.x removed (DataFrame.append, used by gatk/bqsr.py:355; aliased to pd.concat).  The
reference only touches pysam on this path through FastxFile iteration
(recalibrate.py:56,141) and FastxRecord attributes (.name, .sequence, .quality,
.get_quality_array()), so a few lines of stand-in reader are enough.  The
reference's arithmetic (NumPy / SciPy calls) runs untouched.
"""
import sys
import types

import numpy as np


# ---------------------------------------------------------------------------
# Stand-ins for pysam's alignment-side classes, reading the TEXT formats (SAM, FASTA,
# VCF, BED).  Only the attributes the reference touches on the benchmark path exist.
# Also used by oracle/oracle_benchmark.py (both are test infrastructure).
# ---------------------------------------------------------------------------
_CIGAR_OPS = 'MIDNSHP=X'


class AlignedSegment:
    def __init__(self, line):
        f = line.rstrip('\n').split('\t')
        self.query_name = f[0]
        self.flag = int(f[1])
        self.reference_name = f[2]
        self.reference_start = int(f[3]) - 1
        self.mapping_quality = int(f[4])
        self.cigarstring = f[5]
        self.cigartuples = []
        num = ''
        for ch in f[5]:
            if ch.isdigit():
                num += ch
            else:
                self.cigartuples.append((_CIGAR_OPS.index(ch), int(num))); num = ''
        self.query_sequence = f[9]
        self.query_qualities = [ord(c) - 33 for c in f[10]] if f[10] != '*' else None
        self.tags = {}
        for t in f[11:]:
            k, ty, v = t.split(':', 2)
            self.tags[k] = int(v) if ty == 'i' else v
        self.query_length = len(self.query_sequence)
        self.is_paired = bool(self.flag & 1)
        self.is_unmapped = bool(self.flag & 4)
        self.mate_is_unmapped = bool(self.flag & 8)
        self.is_reverse = bool(self.flag & 16)
        self.mate_is_reverse = bool(self.flag & 32)
        self.is_read1 = bool(self.flag & 64)
        self.is_read2 = bool(self.flag & 128)
        self.next_reference_start = int(f[7]) - 1
        self.tlen = self.template_length = int(f[8])
        self._line = line

    # pysam derives these from the CIGAR on every access (the reference's tests reassign .cigartuples)
    @property
    def reference_end(self):
        return self.reference_start + sum(l for op, l in self.cigartuples if op in (0, 2, 3, 7, 8))

    @property
    def query_alignment_start(self):
        n = 0
        for op, l in self.cigartuples:
            if op == 4:
                n += l
            elif op != 5:
                break
        return n

    @property
    def query_alignment_end(self):
        n = 0
        for op, l in reversed(self.cigartuples):
            if op == 4:
                n += l
            elif op != 5:
                break
        return self.query_length - n

    @property
    def query_alignment_length(self):
        return self.query_alignment_end - self.query_alignment_start

    def get_aligned_pairs(self):
        out, q, r = [], 0, self.reference_start
        for op, l in self.cigartuples:
            if op in (0, 7, 8):
                out.extend((q + i, r + i) for i in range(l)); q += l; r += l
            elif op in (1, 4):
                out.extend((q + i, None) for i in range(l)); q += l
            elif op in (2, 3):
                out.extend((None, r + i) for i in range(l)); r += l
        return out

    @classmethod
    def fromstring(cls, line, header=None):
        return cls(line)

    def get_tag(self, k):
        return self.tags[k]

    def has_tag(self, k):
        return k in self.tags

    def set_tag(self, k, v):
        self.tags[k] = v

    def __str__(self):
        return self._line.rstrip('\n')


class AlignmentFile:
    """SAM text reader (the reference's fixtures are the SAM-spec example; BAM needs htslib)."""

    def __init__(self, path, mode='r'):
        self.header_lines, self._reads = [], []
        with open(path) as fh:
            for line in fh:
                if line.startswith('@'):
                    self.header_lines.append(line.rstrip('\n'))
                elif line.strip():
                    self._reads.append(AlignedSegment(line))
        self.header = self
        self._it = iter(self._reads)

    def __iter__(self):
        return iter(self._reads)

    def __next__(self):
        return next(self._it)

    def as_dict(self):
        d = {}
        for line in self.header_lines:
            f = line.split('\t')
            d.setdefault(f[0][1:], []).append({x[:2]: x[3:] for x in f[1:]})
        return d

    def get_index_statistics(self):
        import collections
        return [collections.namedtuple('IndexStats', 'contig mapped unmapped total')('*', len(self._reads), 0, len(self._reads))]


class FastaFile:
    def __init__(self, path):
        self._seqs = {}
        name = None
        with open(path) as fh:
            for line in fh:
                line = line.rstrip('\n')
                if line.startswith('>'):
                    name = line[1:].split()[0]; self._seqs[name] = []
                elif name is not None:
                    self._seqs[name].append(line)
        self._seqs = {k: ''.join(v) for k, v in self._seqs.items()}
        self.references = list(self._seqs)

    def fetch(self, reference=None):
        return self._seqs[reference]


class _VcfRecord:
    def __init__(self, chrom, start, stop):
        self.chrom, self.start, self.stop = chrom, start, stop


class VariantFile:
    def __init__(self, path):
        import gzip
        op = gzip.open if str(path).endswith('.gz') else open
        self._recs = []
        with op(path, 'rt') as fh:
            for line in fh:
                if line.startswith('#') or not line.strip():
                    continue
                f = line.split('\t')
                start = int(f[1]) - 1
                self._recs.append(_VcfRecord(f[0], start, start + len(f[3])))

    def __iter__(self):
        return iter(self._recs)


class _BedRecord:
    def __init__(self, contig, start, end):
        self.contig, self.start, self.end = contig, start, end


def asBed():
    return 'bed'


def tabix_iterator(fh, parser=None):
    for line in fh:
        if line.strip() and not line.startswith(('#', 'track', 'browser')):
            f = line.split()
            yield _BedRecord(f[0], int(f[1]), int(f[2]))


def install(reference_root='/root/reference'):
    for alias, target in (('int', int), ('float', float), ('bool', bool), ('object', object),
                          ('unicode', np.str_), ('NINF', -np.inf)):
        if not hasattr(np, alias):
            setattr(np, alias, target)

    import pandas as pd
    if not hasattr(pd.DataFrame, 'append'):
        pd.DataFrame.append = lambda self, other: pd.concat([self, other])

    class FastxRecord:
        def __init__(self, name=None, sequence=None, quality=None, comment=None):
            self.name, self.sequence, self.quality, self.comment = name, sequence, quality, comment

        def get_quality_array(self, offset=33):
            return [ord(c) - offset for c in self.quality]

        def __str__(self):
            return '@%s\n%s\n+\n%s' % (self.name, self.sequence, self.quality)

    class FastxFile:
        def __init__(self, path):
            self._fh = open(path, 'r')

        def __enter__(self):
            return self

        def __exit__(self, *a):
            self._fh.close()

        def __iter__(self):
            return self

        def __next__(self):
            head = self._fh.readline()
            if not head:
                raise StopIteration
            seq = self._fh.readline().rstrip('\n')
            self._fh.readline()
            qual = self._fh.readline().rstrip('\n')
            fields = head.rstrip('\n')[1:].split(None, 1)
            name = fields[0] if fields else ''
            comment = fields[1] if len(fields) > 1 else None
            return FastxRecord(name, seq, qual, comment)

    pysam = types.ModuleType('pysam')
    pysam.FastxRecord = FastxRecord
    pysam.FastxFile = FastxFile
    # text-format stand-ins for the alignment / variant / reference readers used by the
    # benchmark path (benchmark.py:9-30,57-74,145-164; compare_reads.py:84-139)
    pysam.AlignedSegment = AlignedSegment
    pysam.AlignmentFile = AlignmentFile
    pysam.FastaFile = FastaFile
    pysam.VariantFile = VariantFile
    pysam.tabix_iterator = tabix_iterator
    pysam.asBed = asBed
    sys.modules['pysam'] = pysam
    sys.modules['khmer'] = types.ModuleType('khmer')
    sys.modules['seaborn'] = types.ModuleType('seaborn')
    sys.dont_write_bytecode = True
    if reference_root not in sys.path:
        sys.path.insert(0, reference_root)
