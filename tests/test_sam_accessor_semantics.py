"""
What pins the alignment accessors the §8(f) #1 / #4 goldens depend on.  The goldens of those rows were made by running the
unmodified reference over stand-ins for pysam's AlignedSegment (oracle/_shim.py: pysam is absent from the build container), so
they pin the reference's arithmetic GIVEN those accessors; the accessors themselves -- reference_end, query_alignment_start /
_end / _length, get_aligned_pairs() -- are pinned here, independently of both implementations, by answers worked out BY HAND
from the SAM specification (section 1.4.6: which operations consume the query, which the reference) and pysam's documented
behaviour (soft clips appear in get_aligned_pairs() with reference position None; hard clips and padding nowhere; the
alignment start / end exclude soft clips and ignore hard clips; a read that is ALL clips is left out: what pysam returns for it cannot be
checked here and no aligner emits one).  Both the stand-in and the product's reader
(kbbq.aln.AlignedRead, and the array form of csrc/sam_host.cpp behind AlignmentFile.batch()) must give them.
The reference's own inline answers for the SAM-specification example reads are in tests/test_oracle_benchmark.py.
"""
import numpy as np
import pytest

# (CIGAR, POS (1-based), sequence length) -> reference_end, query_alignment_start, query_alignment_end, aligned pairs
# worked out by hand: M = X consume both, I S consume the query only, D N the reference only, H P nothing.
CASES = [
    ('8M', 5, 8, dict(ref_end=12, qs=0, qe=8, pairs=[(i, 4 + i) for i in range(8)])),
    ('3S5M', 10, 8, dict(ref_end=14, qs=3, qe=8, pairs=[(0, None), (1, None), (2, None)] + [(3 + i, 9 + i) for i in range(5)])),
    ('5M3S', 10, 8, dict(ref_end=14, qs=0, qe=5, pairs=[(i, 9 + i) for i in range(5)] + [(5, None), (6, None), (7, None)])),
    ('2H3S4M1S2H', 1, 8, dict(ref_end=4, qs=3, qe=7, pairs=[(0, None), (1, None), (2, None), (3, 0), (4, 1), (5, 2), (6, 3), (7, None)])),
    ('3M2I3M', 7, 8, dict(ref_end=12, qs=0, qe=8, pairs=[(0, 6), (1, 7), (2, 8), (3, None), (4, None), (5, 9), (6, 10), (7, 11)])),
    ('3M2D3M', 7, 6, dict(ref_end=14, qs=0, qe=6, pairs=[(0, 6), (1, 7), (2, 8), (None, 9), (None, 10), (3, 11), (4, 12), (5, 13)])),
    ('2M100N2M', 3, 4, dict(ref_end=106, qs=0, qe=4, pairs=[(0, 2), (1, 3)] + [(None, 4 + i) for i in range(100)] + [(2, 104), (3, 105)])),
    ('2=1X2=', 20, 5, dict(ref_end=24, qs=0, qe=5, pairs=[(i, 19 + i) for i in range(5)])),
    ('2M1P1I2M', 4, 5, dict(ref_end=7, qs=0, qe=5, pairs=[(0, 3), (1, 4), (2, None), (3, 5), (4, 6)])),
    ('1S2M1I1M1D2M1S', 50, 8, dict(ref_end=55, qs=1, qe=7, pairs=[(0, None), (1, 49), (2, 50), (3, None), (4, 51), (None, 52), (5, 53), (6, 54), (7, None)])),
]


def _line(cigar, pos, n, flag=0):
    return 'r\t%d\tc\t%d\t60\t%s\t*\t0\t0\t%s\t%s\tRG:Z:a\tOQ:Z:%s' % (flag, pos, cigar, 'ACGT' * 30 and ('ACGTACGTACGT' * 20)[:n], 'I' * n, 'J' * n)


@pytest.mark.parametrize('cigar,pos,n,want', CASES)
def test_stand_in_and_product_give_the_specifications_answers(cigar, pos, n, want):
    import _shim
    from kbbq import aln
    for cls in (_shim.AlignedSegment, aln.AlignedRead):
        r = cls(_line(cigar, pos, n))
        assert r.reference_start == pos - 1 and r.query_length == n
        assert r.reference_end == want['ref_end'], cls
        assert (r.query_alignment_start, r.query_alignment_end) == (want['qs'], want['qe']), cls
        assert r.query_alignment_length == want['qe'] - want['qs']
        assert r.get_aligned_pairs() == want['pairs'], cls


def test_the_native_readers_arrays_give_the_same_answers(tmp_path):
    """The array form the kernels are fed from (csrc/sam_host.cpp through AlignmentFile.batch()): reference span and clip ends."""
    from kbbq import aln
    p = tmp_path / 'cases.sam'
    p.write_text('@HD\tVN:1.6\n@SQ\tSN:c\tLN:1000\n@RG\tID:a\tPU:u\n' + ''.join(_line(c, pos, n) + '\n' for c, pos, n, _ in CASES))
    b = aln.AlignmentFile(str(p)).batch()
    assert len(b.flag) == len(CASES)
    for i, (cigar, pos, n, want) in enumerate(CASES):
        assert int(b.pos[i]) == pos - 1 and int(b.qlen[i]) == n, cigar
        assert int(b.ref_span[i]) == want['ref_end'] - (pos - 1), cigar
        assert (int(b.clip[i]) & 0xFFFF, int(b.clip[i]) >> 16) == (want['qs'], want['qe']), cigar
