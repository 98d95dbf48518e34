"""
The benchmark-path oracle (oracle/oracle_benchmark.py) against the reference's own known
answers on the SAM-spec example (reference tests/conftest.py:46-114, tests/test_compare_reads.py:
87-122, tests/test_benchmark.py:7-157, restated as data) and against goldens produced by the
unmodified reference on synthetic truth sets (oracle/gen_golden.py).  CPU only.
"""
import numpy as np
import pytest

from conftest import load_golden

SIMPLE_FA = ">ref\nAGCATGTTAGATAAGATAGCTGTGCTAGTAGGCAGTCAGCGCCAT\n"
SIMPLE_SAM = ("@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:ref\tLN:45\n"
              "r001\t99\tref\t7\t30\t8M2I4M1D3M\t=\t37\t39\tTTAGATAAAGGATACTG\t==99=?<*+/5:@A99:\n"
              "r001\t147\tref\t37\t30\t9M\t=\t7\t-39\tCAGCGGCAT\t><>???>>>\tNM:i:1\n")
SIMPLE_VCF = ('##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tsyndip\n'
              "ref\t10\t.\tG\tT\t30\t.\t.\tGT:AD\t0|1:1,1\n")
SIMPLE_BED = 'ref\t8\t46\n'
# what `samtools fastq -t -N -O` + tr gives for the two reads (the second is reverse-strand)
SIMPLE_FQ = "@r001/1\nTTAGATAAAGGATACTG\n+\n==99=?<*+/5:@A99:\n@r001/2\nATGCCGCTG\n+\n>>>???><>\n"
CORRECT_BENCHMARK = ("9\t42\ttest\t1\n10\t42\ttest\t1\n14\t42\ttest\t1\n20\t42\ttest\t1\n24\t42\ttest\t3\n"
                     "25\t42\ttest\t2\n27\t42\ttest\t2\n28\t42\ttest\t1\n29\t42\ttest\t5\n30\t6\ttest\t4\n"
                     "31\t42\ttest\t1\n32\t42\ttest\t1\n")


@pytest.fixture()
def simple(tmp_path):
    p = {}
    for k, (name, text) in dict(fa=('s.fa', SIMPLE_FA), sam=('s.sam', SIMPLE_SAM), vcf=('s.vcf', SIMPLE_VCF),
                                bed=('s.bed', SIMPLE_BED), fq=('s.fq', SIMPLE_FQ)).items():
        f = tmp_path / name; f.write_text(text); p[k] = str(f)
    return p


@pytest.fixture(scope='module')
def OB(oracle):
    import oracle_benchmark
    return oracle_benchmark


def test_simple_known_answers(OB, simple):
    import _shim
    ref = OB.get_ref_dict(simple['fa'])
    assert bytes(ref['ref']) == b'AGCATGTTAGATAAGATAGCTGTGCTAGTAGGCAGTCAGCGCCAT'
    var = OB.get_var_sites(simple['vcf'])
    assert var == {'ref': [9]}
    full = OB.get_full_skips(ref, var, simple['bed'])
    want = np.zeros(45, dtype=bool); want[0:8] = True; want[9] = True
    assert np.array_equal(full['ref'], want)
    reads = list(_shim.AlignmentFile(simple['sam']))
    assert OB.bam_readname(reads[0]) == 'r001/1' and OB.bam_readname(reads[1]) == 'r001/2'
    assert OB.fastq_readname('r001/2_RG:Z:x') == 'r001/2'
    r1skips = np.zeros(17, dtype=bool); r1skips[3] = True; r1skips[0:2] = True
    r2errs = np.zeros(9, dtype=bool); r2errs[5] = True
    e, s = OB.find_read_errors(reads[0], ref, full)
    assert not e.any() and np.array_equal(s, r1skips)
    e, s = OB.find_read_errors(reads[1], ref, full)
    assert np.array_equal(e, r2errs) and not s.any()
    clipped = _shim.AlignedSegment('clipped\t0\tref\t9\t255\t1M9H\t*\t0\t0\tA\t)')
    e, s = OB.find_read_errors(clipped, ref, full)
    assert list(e) == [False] and list(s) == [False]
    clipped.cigartuples = [('L', 9)]
    with pytest.raises(ValueError):
        OB.find_read_errors(clipped, ref, full)
    ed = OB.get_error_dict(reads, ref, full)
    assert np.array_equal(ed['r001/2'][0], np.flip(r2errs)) and np.array_equal(ed['r001/1'][1], r1skips)
    a, t = OB.calculate_q(np.array([False, True, True] + [False] * 100), np.array([3, 2, 1] + [1] * 100))
    assert list(a) == [0, 20, 0, 42] and list(t) == [0, 101, 1, 1]
    assert OB.format_benchmark(a, 'test', t) == "1\t20\ttest\t101\n2\t0\ttest\t1\n3\t42\ttest\t1\n"
    a, t = OB.benchmark_bam(reads, ref, var, bed_path=simple['bed'])
    assert OB.format_benchmark(a, 'test', t) == CORRECT_BENCHMARK
    a, t = OB.benchmark_fastq(simple['fq'], reads, ref, var, simple['bed'])
    assert OB.format_benchmark(a, 'test', t) == CORRECT_BENCHMARK


@pytest.mark.parametrize('name', ['bench_a', 'bench_b'])
def test_matches_reference_goldens(OB, oracle, name, tmp_path):
    import _shim
    info, gold = load_golden(name)
    paths = OB.synth_truthset(str(tmp_path), **info['case'])
    assert {k: oracle.sha256(open(v, 'rb').read()) for k, v in paths.items()} == info['input_sha256']
    ref, var = OB.get_ref_dict(paths['fa']), OB.get_var_sites(paths['vcf'])
    full = OB.get_full_skips(ref, var, paths['bed'])
    reads = list(_shim.AlignmentFile(paths['sam']))
    ed = OB.get_error_dict(reads, ref, full)
    assert list(ed) == info['read_keys']
    assert np.array_equal(np.concatenate([ed[k][0] for k in ed]).astype(np.uint8), gold['errors'])
    assert np.array_equal(np.concatenate([ed[k][1] for k in ed]).astype(np.uint8), gold['skips'])
    for tag, (a, t) in dict(bam=OB.benchmark_bam(reads, ref, var, bed_path=paths['bed']),
                            bam_oq=OB.benchmark_bam(reads, ref, var, use_oq=True, bed_path=paths['bed']),
                            fastq=OB.benchmark_fastq(paths['fq'], reads, ref, var, paths['bed']),
                            nobed=OB.benchmark_bam(reads, ref, var)).items():
        assert np.array_equal(a, gold[tag + '_q']) and np.array_equal(t, gold[tag + '_n']), tag
    a, t = OB.benchmark_bam(reads, ref, var, bed_path=paths['bed'])
    assert OB.format_benchmark(a, 'lbl', t) == info['printed']['bam']
    a, t = OB.benchmark_fastq(paths['fq'], reads, ref, var, paths['bed'])
    assert OB.format_benchmark(a, 'lbl', t) == info['printed']['fastq']
