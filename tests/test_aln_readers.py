"""CPU tests of the product's text readers (kbbq/aln.py) against the oracle-side stand-ins."""
import numpy as np
import pytest

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


ARRAYS = ('flag', 'contig', 'pos', 'pnext', 'tlen', 'qlen', 'ref_span', 'clip', 'cig_n', 'rg', 'qual_len', 'oq_len', 'cigar')


def _same_batches(a, b):
    assert a.n == b.n and a.contig_names == b.contig_names and a.rg_ids == b.rg_ids
    for name in ARRAYS:
        assert np.array_equal(getattr(a, name), getattr(b, name)), name
    pitch = (max(a.maxlen, 1) + 15) // 16 * 16
    for which in (0, 1, 2):
        assert np.array_equal(a.plane(which, pitch), b.plane(which, pitch)), which


def test_bam_and_gzipped_sam_are_read_like_sam_text(OB, oracle, tmp_path):
    """The reference reads BAM through pysam; here BAM (BGZF blocks inflated in parallel, records rendered as SAM
    lines by csrc/bam_host.cpp) and gzip-compressed SAM go through the same parser as SAM text: identical arrays,
    planes, names and -- for BAM -- identical rendered lines, on the synthetic truth set and the BAM-sourced-tally set."""
    import gzip
    import bamwriter
    import oracle_bqsr as OQ
    sets = [OB.synth_truthset(str(tmp_path / 'bench'), seed=5, npairs=600)['sam'] if (tmp_path / 'bench').mkdir() is None else None,
            OQ.synth_bqsr_set(str(tmp_path), seed=9, npairs=500, S=75)['sam']]
    for k, sam in enumerate(sets):
        text = open(sam).read()
        ref = aln.AlignmentFile(sam)
        bam = aln.AlignmentFile(bamwriter.write_bam(tmp_path / ('x%d.bam' % k), text))
        _same_batches(ref.batch(), bam.batch())
        assert [bam.batch().line(i) for i in range(ref.batch().n)] == [ln for ln in text.split('\n') if ln and not ln.startswith('@')]
        assert bam.header.as_dict() == ref.header.as_dict() and bam.batch().names() == ref.batch().names()
        gz = tmp_path / ('x%d.sam.gz' % k)
        with gzip.open(gz, 'wt') as fh:
            fh.write(text)
        _same_batches(ref.batch(), aln.AlignmentFile(str(gz)).batch())
        # the read objects built on demand are the same objects' worth of attributes
        for r, q in list(zip(ref, bam))[:40]:
            assert (r.query_name, r.flag, r.reference_start, r.cigartuples, r.query_sequence) == \
                   (q.query_name, q.flag, q.reference_start, q.cigartuples, q.query_sequence)
            assert list(r.query_qualities) == list(q.query_qualities) and r.get_tag('RG') == q.get_tag('RG')


def test_bam_edge_cases_and_malformed_files(tmp_path):
    import bamwriter
    hdr = '@HD\tVN:1.6\n@SQ\tSN:c1\tLN:1000\n@SQ\tSN:c2\tLN:500\n@RG\tID:a\tPU:u\n'
    recs = ['r1\t99\tc1\t5\t60\t2S4M1I3M2D1M\t=\t40\t50\tACGTNACGTAA\tIIIIIIIIIII\tRG:Z:a\tOQ:Z:JJJJJJJJJJJ\tNM:i:3\tXA:A:q\tXF:f:1.5\tXB:B:c,-1,2,3\tXS:B:S,1,65535\tXH:H:1AE3',
            'r2\t4\t*\t0\t0\t*\t*\t0\t0\t*\t*',
            'r3\t16\tc2\t7\t3\t5M\tc1\t100\t-20\tAC=TM\t*\tXI:i:-70000\tXU:i:4000000000',
            'r4\t0\tc2\t1\t255\t3M\t*\t0\t0\tGGG\t!!~']
    text = hdr + '\n'.join(recs) + '\n'
    sam = tmp_path / 'e.sam'; sam.write_text(text)
    bam = aln.AlignmentFile(bamwriter.write_bam(tmp_path / 'e.bam', text))
    _same_batches(aln.AlignmentFile(str(sam)).batch(), bam.batch())
    assert [bam.batch().line(i) for i in range(4)] == recs
    # no header text in the BAM: the binary reference list becomes @SQ lines
    bare = aln.AlignmentFile(bamwriter.write_bam(tmp_path / 'bare.bam', text, header_text=False))
    assert bare.batch().contig_names == bam.batch().contig_names and [x['SN'] for x in bare.header.as_dict()['SQ']] == ['c1', 'c2']
    assert bare.batch().rg_ids == [] and int(bare.batch().rg[0]) < 0
    # an uncompressed BAM image, an empty BAM, and many blocks
    raw = bamwriter.sam_to_bam_bytes(text)
    (tmp_path / 'raw.bam').write_bytes(raw)
    assert aln.AlignmentFile(str(tmp_path / 'raw.bam')).batch().n == 4
    assert aln.AlignmentFile(bamwriter.write_bam(tmp_path / 'none.bam', hdr)).batch().n == 0
    (tmp_path / 'small.bam').write_bytes(bamwriter.bgzf(raw, block=37))
    _same_batches(bam.batch(), aln.AlignmentFile(str(tmp_path / 'small.bam')).batch())
    # damaged files: an error, never a crash
    z = bamwriter.bgzf(raw)
    for name, data in (('cut', z[:len(z) // 2]), ('crc', z[:40] + bytes([z[40] ^ 1]) + z[41:]), ('rec', bamwriter.bgzf(raw[:-7])),
                       ('magic', bamwriter.bgzf(b'BAM\1' + b'\xff' * 40)), ('refs', bamwriter.bgzf(raw[:len(hdr) + 14]))):
        (tmp_path / (name + '.bam')).write_bytes(data)
        with pytest.raises(ValueError):
            aln.AlignmentFile(str(tmp_path / (name + '.bam')))


SIMPLE_SAM = ("@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:ref\tLN:45\n"
              "r001\t99\tref\t7\t30\t8M2I4M1D3M\t=\t37\t39\tTTAGATAAAGGATACTG\t==99=?<*+/5:@A99:\n"
              "r001\t147\tref\t37\t30\t9M\t=\t7\t-39\tCAGCGGCAT\t><>???>>>\tNM:i:1\n")


def test_readdata_bam_factories(tmp_path):
    """The reference's tests/test_read.py:21-43,65-73 (ReadData.from_bamread / load_rgs_from_bamfile), on reads that
    come out of a BAM file of its SAM-spec example (reference tests/conftest.py:60-81)."""
    import bamwriter
    from kbbq import read
    read.ReadData.rg_to_pu = dict(); read.ReadData.rg_to_int = dict(); read.ReadData.numrgs = 0
    try:
        read.ReadData(seq=np.array(['A', 'T', 'G']), qual=np.array([6, 10, 3]), skips=np.array([False, False, True]), name='read01',
                      rg=0, second=False, errors=np.array([False, True, True]))          # the reference's conftest registers rg 0 first
        read.ReadData.rg_to_pu = dict(); read.ReadData.rg_to_int = dict(); read.ReadData.numrgs = 0
        reads = list(aln.AlignmentFile(bamwriter.write_bam(tmp_path / 'simple.bam', SIMPLE_SAM)))
        bamread = reads[0]
        want = np.array([ord(c) - 33 for c in '==99=?<*+/5:@A99:'])
        r = read.ReadData.from_bamread(bamread)
        assert np.array_equal(r.qual, want) and r.rg is None and not r.second and r.name == 'r001'
        assert not r.skips.any() and not r.errors.any() and ''.join(r.seq) == 'TTAGATAAAGGATACTG'
        bamread.set_tag('OQ', '(' * 17)
        bamread.set_tag('RG', 'foo')
        r = read.ReadData.from_bamread(bamread, use_oq=True)
        assert np.array_equal(r.qual, np.array([7] * 17)) and r.rg == 'foo'
        assert read.ReadData.rg_to_int[None] == 0 and read.ReadData.rg_to_int['foo'] == 1 and read.ReadData.numrgs == 2
        bamread.is_reverse = True
        r = read.ReadData.from_bamread(bamread)
        assert np.array_equal(r.qual, np.flip(want)) and np.array_equal(r.seq, np.array(list('CAGTATCCTTTATCTAA')))
        r2 = read.ReadData.from_bamread(reads[1])                    # flag 147: reverse strand, second in pair
        assert r2.second and ''.join(r2.seq) == 'ATGCCGCTG' and list(r2.qual) == [ord(c) - 33 for c in reversed('><>???>>>')]
        read.ReadData.rg_to_pu = dict(); read.ReadData.rg_to_int = dict(); read.ReadData.numrgs = 0
        hdr = '@HD\tVN:1.6\n@SQ\tSN:ref\tLN:45\n@RG\tID:FOO\tSM:BAR\tPU:BAZ\n'
        read.ReadData.load_rgs_from_bamfile(aln.AlignmentFile(bamwriter.write_bam(tmp_path / 'rg.bam', hdr)))
        assert read.ReadData.rg_to_int['FOO'] == 0 and read.ReadData.rg_to_pu['FOO'] == 'BAZ' and read.ReadData.numrgs == 1
        assert read.bamread_get_quals(reads[1]).tolist() == [ord(c) - 33 for c in '><>???>>>']
    finally:
        read.ReadData.rg_to_pu = dict(); read.ReadData.rg_to_int = dict(); read.ReadData.numrgs = 0


def test_small_helpers_of_compare_reads(tmp_path):
    """tstamp / load_positions / get_var_sites (reference compare_reads.py:26-68)."""
    import re
    from kbbq import benchmark, compare_reads
    assert re.fullmatch(r'\[ \d{4}-\d\d-\d\d \d\d:\d\d:\d\d \]', compare_reads.tstamp())
    bed = tmp_path / 'p.bed'; bed.write_text('ref\t8\t11\nother\t0\t2\nref\t20\t21\n')
    assert compare_reads.load_positions(str(bed)) == {'ref': [8, 9, 10, 20], 'other': [0, 1]}
    vcf = tmp_path / 's.vcf'
    vcf.write_text('##fileformat=VCFv4.2\n##contig=<ID=ref,length=45>\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n'
                   'ref\t10\t.\tG\tT\t30\t.\t.\nref\t20\t.\tACG\tA\t30\t.\t.\n')
    assert compare_reads.get_var_sites(str(vcf)) == benchmark.get_var_sites(str(vcf)) == {'ref': [9, 19, 20, 21]}


def test_fasta_reader_layouts(tmp_path):
    """aln.FastaFile on odd but legal layouts: text before the first header, '>' inside a header's description,
    blank lines, CRLF, an empty record, no trailing newline; plain and gzip-compressed."""
    import gzip
    cases = (('junk before\n>c1 some >desc\nACGT\nAC\n\n>c2\r\nGG\r\nTT\r\n>empty\n>c3\nA', {'c1': 'ACGTAC', 'c2': 'GGTT', 'empty': '', 'c3': 'A'}),
             ('>only', {'only': ''}), ('', {}), ('no header\nACGT\n', {}), ('>a b\n>c\nT\n', {'a': '', 'c': 'T'}))
    for k, (text, want) in enumerate(cases):
        p = tmp_path / ('f%d.fa' % k); p.write_text(text)
        f = aln.FastaFile(str(p))
        assert f.references == list(want) and {r: f.fetch(r) for r in f.references} == want
        with gzip.open(str(p) + '.gz', 'wt') as fh:
            fh.write(text)
        assert aln.FastaFile(str(p) + '.gz')._seqs == want


def test_bam_records_round_trip_property(tmp_path):
    """Hypothesis: arbitrary alignment records (names, flags, positions, CIGARs, IUPAC sequences, qualities or '*',
    every tag type incl. B arrays) written as BAM by the test-side writer come back from the library's BAM reader as
    exactly the SAM lines they were made from."""
    import bamwriter
    from hypothesis import given, settings, strategies as st, HealthCheck
    name = st.text(alphabet='abcXYZ019_./:-', min_size=1, max_size=20)
    cigar_ops = st.lists(st.tuples(st.integers(1, 300), st.sampled_from('MIDNSHP=X')), min_size=0, max_size=6)
    small = st.integers(-128, 127) | st.integers(0, 65535) | st.integers(-2 ** 31, 2 ** 32 - 1)
    tag_value = st.one_of(
        st.tuples(st.just('A'), st.sampled_from('qZ!~')),
        st.tuples(st.just('i'), small.map(str)),
        st.tuples(st.just('Z'), st.text(alphabet='ACGT!#IJ~ x:;', min_size=0, max_size=12)),
        st.tuples(st.just('H'), st.text(alphabet='0123456789ABCDEF', min_size=0, max_size=8).map(lambda s: s[:len(s) // 2 * 2])),
        st.tuples(st.just('f'), st.sampled_from(['1.5', '0', '-2.25', '1e+10', '3.40282e+38'])),
        st.tuples(st.just('B'), st.tuples(st.sampled_from('cCsSiI'), st.lists(st.integers(0, 100), max_size=5)).map(
            lambda t: t[0] + ''.join(',%d' % v for v in t[1]))))
    tags = st.lists(st.tuples(st.sampled_from(['NM', 'XA', 'XB', 'OQ', 'RG', 'ZZ', 'a1']), tag_value), max_size=4, unique_by=lambda t: t[0])

    @st.composite
    def record(draw):
        seq = draw(st.text(alphabet='=ACMGRSVTWYHKDBN', min_size=0, max_size=40))
        qual = '*' if not seq or draw(st.booleans()) else ''.join(draw(st.lists(st.sampled_from('!#5AIJ~'), min_size=len(seq), max_size=len(seq))))
        ops = draw(cigar_ops)
        ref = draw(st.sampled_from(['*', 'c1', 'c2']))
        nxt = draw(st.sampled_from(['*', '=', 'c1', 'c2']))
        if ref == '*' and nxt == '=':
            nxt = '*'
        if nxt == ref and ref != '*':
            nxt = '='                                                # samtools prints '=' for the same reference
        f = [draw(name), str(draw(st.integers(0, 4095))), ref, str(draw(st.integers(0, 2 ** 29))), str(draw(st.integers(0, 255))),
             ''.join('%d%s' % o for o in ops) or '*', nxt, str(draw(st.integers(0, 2 ** 29))), str(draw(st.integers(-2 ** 30, 2 ** 30))),
             seq or '*', qual]
        f += ['%s:%s:%s' % (k, t, v) for k, (t, v) in draw(tags)]
        return '\t'.join(f)
    hdr = '@HD\tVN:1.6\n@SQ\tSN:c1\tLN:1000000000\n@SQ\tSN:c2\tLN:500\n@RG\tID:a\tPU:u\n'
    counter = [0]

    @settings(max_examples=60, deadline=None, suppress_health_check=list(HealthCheck))
    @given(st.lists(record(), min_size=0, max_size=8))
    def check(recs):
        counter[0] += 1
        bam = bamwriter.write_bam(tmp_path / ('h%d.bam' % counter[0]), hdr + ''.join(r + '\n' for r in recs))
        try:
            b = aln.AlignmentFile(bam).batch()
        except ValueError:
            # the SAM parser refuses some legal-in-BAM records (e.g. a CIGAR whose query length disagrees): not this test's subject
            return
        assert [b.line(i) for i in range(b.n)] == recs
    check()


def test_compressed_fasta_vcf_bed_come_through_the_librarys_inflater(tmp_path):
    """aln._read_bytes (kbbq_text_open): .gz FASTA / VCF / BED -- plain gzip and bgzip -- read like the plain files; a missing file is
    open()'s FileNotFoundError, a damaged one a ValueError."""
    import gzip
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import bamwriter
    from kbbq import aln
    fa = b'>c1 first\nACGTNNAC\nGT\n>c2\n\nTTTT\r\nGG\n'
    vcf = b'##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\n' + b''.join(b'c1\t%d\t.\tAC\tA\n' % i for i in range(1, 3000))
    bed = b'track x\nc1 1 5\nc2\t0\t3\n'
    (tmp_path / 'g.fa').write_bytes(fa)
    plain = aln.FastaFile(str(tmp_path / 'g.fa'))
    for tag, pack in (('gz', lambda b: gzip.compress(b, 6)), ('bgz', lambda b: bamwriter.bgzf(b, block=700))):
        for name, blob in (('g.fa', fa), ('v.vcf', vcf), ('b.bed', bed)):
            (tmp_path / ('%s.%s.gz' % (name, tag))).write_bytes(pack(blob))
        z = aln.FastaFile(str(tmp_path / ('g.fa.%s.gz' % tag)))
        assert z.references == plain.references == ['c1', 'c2'] and all(z.fetch(r) == plain.fetch(r) for r in plain.references)
        recs = list(aln.read_vcf(str(tmp_path / ('v.vcf.%s.gz' % tag))))
        assert len(recs) == 2999 and (recs[0].chrom, recs[0].start, recs[0].stop) == ('c1', 0, 2)
        with aln._open(str(tmp_path / ('b.bed.%s.gz' % tag))) as fh:
            assert [(r.contig, r.start, r.end) for r in aln.read_bed(fh)] == [('c1', 1, 5), ('c2', 0, 3)]
    with pytest.raises(FileNotFoundError):
        aln.FastaFile(str(tmp_path / 'missing.fa.gz'))
    whole = gzip.compress(fa * 2000, 6)
    (tmp_path / 'cut.fa.gz').write_bytes(whole[:len(whole) // 2])
    with pytest.raises(ValueError):
        aln.FastaFile(str(tmp_path / 'cut.fa.gz'))
