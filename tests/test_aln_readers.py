"""CPU tests of the product's text readers (kbbq/aln.py) against the oracle-side stand-ins."""
import numpy as np

from test_oracle_benchmark import simple, OB    # noqa: F401
from kbbq import aln


def test_readers_agree_with_oracle_side(OB, tmp_path):
    import _shim
    paths = OB.synth_truthset(str(tmp_path), seed=5, npairs=40)
    mine, theirs = list(aln.AlignmentFile(paths['sam'])), list(_shim.AlignmentFile(paths['sam']))
    assert len(mine) == len(theirs) == 80
    for a, b in zip(mine, theirs):
        for k in ('query_name', 'flag', 'reference_name', 'reference_start', 'reference_end', 'cigartuples',
                  'query_sequence', 'query_qualities', 'is_reverse', 'is_read2', 'query_length'):
            assert getattr(a, k) == getattr(b, k), k
        assert a.get_tag('OQ') == b.get_tag('OQ') and a.has_tag('RG') and not a.has_tag('XX')
    fa, ofa = aln.FastaFile(paths['fa']), _shim.FastaFile(paths['fa'])
    assert fa.references == ofa.references
    assert all(fa.fetch(reference=c) == ofa.fetch(reference=c) for c in fa.references)
    assert [(r.chrom, r.start, r.stop) for r in aln.read_vcf(paths['vcf'])] == \
        [(r.chrom, r.start, r.stop) for r in _shim.VariantFile(paths['vcf'])]
    with open(paths['bed']) as f1, open(paths['bed']) as f2:
        assert [(r.contig, r.start, r.end) for r in aln.read_bed(f1)] == \
            [(r.contig, r.start, r.end) for r in _shim.tabix_iterator(f2)]
    c = aln.chars('ACGTN')
    assert c.dtype == np.dtype('U1') and list(c) == list('ACGTN') and list(aln.codes(c)) == [65, 67, 71, 84, 78]
    assert aln.parse_cigar('8M2I4M1D3M') == [(0, 8), (1, 2), (0, 4), (2, 1), (0, 3)] and aln.parse_cigar('*') == []
