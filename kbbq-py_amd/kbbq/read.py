"""
kbbq.read -- ReadData, the per-read record CovariateData.consume_read takes
(reference kbbq/read.py:21-378).  Host-side data class: arrays of one read plus
the class-level read-group registry.  The BAM factories (from_bamread,
load_rgs_from_bamfile) take the read / file objects of kbbq.aln (or pysam's).
"""
import numpy as np

from . import compare_reads


class ReadData():
    """Minimal per-read information: seq (array of 1-char strings), qual (int array),
    skips / errors (bool arrays), name, rg (id, registered in first-appearance order) and
    second (second-in-pair flag).  Treat the class attributes as read-only."""

    rg_to_pu = dict()
    rg_to_int = dict()
    numrgs = 0

    def __init__(self, seq, qual, skips, name, rg, second, errors):
        self.seq = seq
        self.qual = qual
        self.skips = skips
        self.name = name
        self.rg = rg
        if rg not in ReadData.rg_to_pu:
            # unseen read group: its platform unit is the id itself
            cls = self.__class__
            cls.rg_to_pu[rg] = rg
            cls.rg_to_int[rg] = ReadData.numrgs
            cls.numrgs = ReadData.numrgs + 1
        self.second = second
        self.errors = errors

    @classmethod
    def from_fastq(cls, fastqread, rg=None, second=None, namedelimiter='_'):
        """Build from a FASTQ record (.name/.sequence/.get_quality_array()).  rg and second are
        inferred from the name when not given: the last 'RG:' field's text after its last ':'
        and a first field ending in '/2'; a trailing /1 or /2 is dropped from the stored name."""
        seq = np.array(list(fastqread.sequence), dtype=np.str_)
        fields = fastqread.name.split(sep=namedelimiter)
        if rg is None:
            ids = [f.split(':')[-1] for f in fields if f[0:3] == 'RG:']
            if ids:
                rg = ids[-1]
        if second is None:
            second = fields[0][-2:] == '/2'
        if fields[0].endswith(('/1', '/2')):
            fields[0] = fields[0][:-2]
        n = len(seq)
        return cls(seq=seq, qual=np.array(fastqread.get_quality_array(), dtype=np.int_),
                   skips=np.zeros(n, dtype=bool), name=fields[0], rg=rg, second=second,
                   errors=np.zeros(n, dtype=bool))

    @classmethod
    def from_bamread(cls, bamread, use_oq=False):
        """From an aligned read (reference read.py:101-141): sequence and qualities in SEQUENCING orientation
        (reverse-strand reads reverse-complemented, letters outside ACGT -> 'N'), the OQ tag's qualities when
        use_oq, rg = the RG tag or None; skips / errors all False."""
        seq = np.array(list(bamread.query_sequence), dtype=np.str_)
        qual = bamread_get_quals(bamread, use_oq)
        if bamread.is_reverse:
            seq = compare_reads.Dinucleotide.veccomplement(np.flip(seq), 'N')
            qual = np.flip(qual)
        n = len(seq)
        return cls(seq=seq, qual=qual, skips=np.zeros(n, dtype=bool), name=bamread.query_name,
                   rg=bamread.get_tag('RG') if bamread.has_tag('RG') else None, second=bamread.is_read2,
                   errors=np.zeros(n, dtype=bool))

    @classmethod
    def load_rgs_from_bamfile(cls, bamfileobj):
        """Register the header's read groups (ID -> PU, ID -> index in header order; reference read.py:199-218)."""
        for rg in bamfileobj.header.as_dict()['RG']:
            cls.rg_to_pu[rg['ID']] = rg['PU']
            cls.rg_to_int[rg['ID']] = cls.numrgs
            cls.numrgs = cls.numrgs + 1

    def str_qual(self, offset=33):
        return [chr(int(q) + offset) for q in self.qual]

    def canonical_name(self):
        return self.name + ('/2' if self.second else '/1')

    def get_rg_int(self):
        return self.__class__.rg_to_int[self.rg]

    def get_pu(self):
        return self.__class__.rg_to_pu[self.rg]

    def not_skipped_errors(self):
        return np.logical_and(self.errors, ~self.skips)

    def get_rg_errors(self):
        rg = np.broadcast_to(self.get_rg_int(), len(self))
        return rg[self.not_skipped_errors()], rg[~self.skips]

    def get_q_errors(self):
        return self.qual[self.not_skipped_errors()], self.qual[~self.skips]

    def get_cycle_array(self):
        return compare_reads.generic_cycle_covariate(len(self), self.second)

    def get_cycle_errors(self):
        cycle = self.get_cycle_array()
        return cycle[self.not_skipped_errors()], cycle[~self.skips]

    def get_dinucleotide_array(self, minscore=6):
        return compare_reads.generic_dinuc_covariate(np.asarray(self.seq, dtype='U1'),
                                                     np.asarray(self.qual), minscore)

    def get_dinuc_errors(self, minscore=6):
        dinuc = self.get_dinucleotide_array(minscore)
        dvalid = np.logical_and(dinuc != -1, ~self.skips)
        return dinuc[np.logical_and(dvalid, self.errors)], dinuc[dvalid]

    def __len__(self):
        return len(self.seq)


def bamread_get_quals(read, use_oq=False):
    """Qualities of an aligned read as an int array: the OQ tag's when use_oq (reference read.py:398-414)."""
    if use_oq:
        return compare_reads.bamread_get_oq(read)
    return np.array(read.query_qualities, dtype=np.int_)


def bamread_get_oq(read):
    """The OQ tag's qualities as an int array (reference read.py:385-396; the same function as compare_reads')."""
    return compare_reads.bamread_get_oq(read)

