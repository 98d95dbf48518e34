"""
The host C++ of libkbbq_hip (FASTQ reader / packer / writer, gammaln pool) under AddressSanitizer and
UndefinedBehaviorSanitizer, on well-formed and malformed inputs.  CPU only (GPU sanitizers are not available on
the pool); the harness is tests/native/host_sanitize.cpp.
"""
import os
import subprocess

import pytest

from conftest import ROOT


@pytest.fixture(scope='module')
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp('san') / 'host_sanitize')
    csrc = os.path.join(ROOT, 'kbbq-py_amd', 'csrc')
    cmd = ['g++', '-std=c++17', '-O1', '-g', '-fsanitize=address,undefined', '-fno-omit-frame-pointer', '-pthread',
           '-o', exe, os.path.join(ROOT, 'tests', 'native', 'host_sanitize.cpp'),
           os.path.join(csrc, 'fastq_host.cpp'), os.path.join(csrc, 'fastq_stream.cpp'), os.path.join(csrc, 'solve_host.cpp'),
           os.path.join(csrc, 'sam_host.cpp'), os.path.join(csrc, 'bam_host.cpp'), os.path.join(csrc, 'parallel_gunzip.cpp'), '-lz', '-ldl']
    subprocess.check_call(cmd)

    def run(*args):
        env = dict(os.environ, ASAN_OPTIONS='detect_leaks=1:abort_on_error=0', UBSAN_OPTIONS='halt_on_error=1:print_stacktrace=1',
                   KBBQ_HOST_THREADS='4', KBBQ_SCAN_CHUNK='97',       # the scan's chunks on threads even for small inputs
                   KBBQ_PGZ_MIN_BYTES='0', KBBQ_PGZ_CHUNK='2000')   # ... and gzip members cut into chunks (csrc/parallel_gunzip.cpp)
        r = subprocess.run([exe] + list(args), capture_output=True, timeout=300, env=env)
        out, err = r.stdout.decode('latin-1'), r.stderr.decode('latin-1')      # error texts may quote bytes of a damaged file
        assert 'ERROR: AddressSanitizer' not in err and 'runtime error' not in err, err[-3000:]
        assert r.returncode == 0, (r.returncode, err[-2000:])
        return out
    return run


def _write(p, text, mode='w'):
    with open(p, mode) as fh:
        fh.write(text)
    return str(p)


def test_wellformed_pairs(harness, oracle, tmp_path):
    n = 5000
    seq, cseq, qual, meta = oracle.synth(0, n, n, 4, 36, 151, 3)
    names = oracle.synth_names(0, n, 3, with_rg=True)
    fa, fb = str(tmp_path / 'a.fq'), str(tmp_path / 'b.fq')
    oracle.write_fastq(fa, names, seq, qual, meta)
    oracle.write_fastq(fb, names, cseq, qual, meta)
    out = harness('pair', fa, fb, '1')
    assert 'scan rc=0 n=%d' % n in out and 'fill rc=0' in out and out.count('rc=0 bytes=') == 2
    assert 'bad=0 negative=' in out and 'found=0' not in out
    assert 'shard rc=0 records=%d/%d' % (2 * n // 3 - n // 3, 2 * n // 3 - n // 3) in out and 'misaligned rc=-4' in out
    assert 'job rc=0 n=%d' % n in out and out.split('job ')[1].split('\n')[0].split(' ', 1)[1].startswith(out.split('scan ')[1].split('\n')[0].split(' ', 1)[1])
    out = harness('pair', fa, '-', '0')
    assert 'scan rc=0' in out
    # the sequential reader over the same pair, in segments of 64 KB and of everything at once: the same file-wide facts
    want = out_pair = harness('pair', fa, fb, '1')
    S = int(want.split('scan rc=0 n=%d S=' % n)[1].split()[0])
    for seg in ('65536', '300000', '1000000000'):
        out = harness('stream', fa, fb, '1', seg)
        assert 'stream records=%d usable=%d' % (n, n) in out and ' S=%d R=3' % S in out and 'offender' not in out, out
    assert 'stream records=%d' % n in harness('stream', fa, '-', '0', '70000')


def test_malformed_inputs_do_not_touch_memory_they_do_not_own(harness, tmp_path):
    cases = {
        'empty': '',
        'no_trailing_newline': '@r1/1\nACGT\n+\nIIII',
        'crlf': '@r1/1 c\r\nACGT\r\n+\r\nIIII\r\n@r2/2\r\nAC\r\n+\r\nII\r\n',
        'three_lines': '@r1\nACGT\n+\n',
        'no_at': 'r1\nACGT\n+\nIIII\n',
        'short_quality': '@r1\nACGT\n+\nII\n',
        'long_quality': '@r1\nAC\n+\nIIIIIIII\n',
        'blank_lines': '@r1\n\n+\n\n@r2\n\n+\n\n',
        'only_newlines': '\n\n\n\n',
        'rg_missing': '@r1/1_RG:Z:a\nACGT\n+\nIIII\n@r2/2\nACGT\n+\nIIII\n',
        'rg_not_rg': '@r1/1_XX:Z:a\nACGT\n+\nIIII\n',
        'binary': '@r\x00\x01\nAC\xff\xfe\n+\n\x00\x7f\x80\xff\n',
        'huge_name': '@' + 'n' * 100000 + '_RG:Z:' + 'g' * 70000 + '\nACGT\n+\nIIII\n',
        # wrapped records (the unwrapping pass of the mapped reader), good and bad
        'wrapped': '@r1/1_RG:Z:a c\nACGT\nAC\n+r1\nII\nIIII\n@r2/2_RG:Z:a\nACGTAC\n+\n@IIII\nI\n',
        'wrapped_short_quality': '@r1\nACGT\nACGT\n+\nIIII\nII\n',
        'wrapped_no_plus': '@r1\nACGT\nACGT\nACGT\n',
        'wrapped_long_quality': '@r1\nAC\nGT\n+\nIIIII\nIII\n',
        'wrapped_ends_in_sequence': '@r1\nACGT\nAC\n+\nIIIIII\n@r2\nAC',
        'wrapped_crlf': '@r1\r\nACGT\r\nAC\r\n+\r\nIII\r\nIII\r\n',
    }
    good = _write(tmp_path / 'good.fq', '@r1/1_RG:Z:a\nACGT\n+\nIIII\n@r2/2_RG:Z:a\nACGT\n+\nIIII\n')
    for name, text in cases.items():
        p = _write(tmp_path / (name + '.fq'), text.encode('latin-1'), 'wb')
        for infer in ('0', '1'):
            harness('pair', p, '-', infer)
            harness('pair', p, good, infer)
            harness('pair', good, p, infer)
            harness('stream', p, '-', infer, '65536')
            harness('stream', p, good, infer, '65536')
            harness('stream', good, p, infer, '65536')
    harness('pair', str(tmp_path / 'does_not_exist.fq'), '-', '0')
    # gzip input, intact / truncated / garbage after the magic
    import gzip
    good_text = open(good, 'rb').read() * 500
    gz = tmp_path / 'good.fq.gz'
    with gzip.open(gz, 'wb') as fh:
        fh.write(good_text)
    assert 'scan rc=0 n=1000' in harness('pair', str(gz), str(gz), '1')
    raw = open(gz, 'rb').read()
    harness('pair', _write(tmp_path / 'cut.fq.gz', raw[:len(raw) // 2], 'wb'), '-', '0')
    harness('pair', _write(tmp_path / 'junk.fq.gz', b'\x1f\x8b' + b'\x00' * 50, 'wb'), '-', '0')
    assert 'stream records=1000 usable=1000' in harness('stream', str(gz), str(gz), '1', '65536')      # inflated as it is read
    harness('stream', str(tmp_path / 'cut.fq.gz'), '-', '0')
    harness('stream', str(tmp_path / 'junk.fq.gz'), '-', '0')
    harness('stream', _write(tmp_path / 'members.fq.gz', gzip.compress(good_text[:30000]) + gzip.compress(good_text[30000:]) + b'\0' * 64, 'wb'), str(gz), '1', '65536')
    import bamwriter
    bz = bamwriter.bgzf(good_text, block=0x1000)
    assert 'stream records=1000 usable=1000' in harness('stream', _write(tmp_path / 'blocks.fq.gz', bz, 'wb'), str(gz), '1', '65536')
    harness('stream', _write(tmp_path / 'blocks_cut.fq.gz', bz[:len(bz) // 2 + 7], 'wb'), '-', '0')
    harness('stream', _write(tmp_path / 'blocks_flip.fq.gz', bz[:len(bz) // 2] + bytes([bz[len(bz) // 2] ^ 0x11]) + bz[len(bz) // 2 + 1:], 'wb'), '-', '0')
    harness('stream', _write(tmp_path / 'blocks_bsize.fq.gz', bz[:16] + b'\xff\xff' + bz[18:], 'wb'), '-', '0')
    harness('stream', _write(tmp_path / 'flip.fq.gz', raw[:len(raw) // 2] + bytes([raw[len(raw) // 2] ^ 0x11]) + raw[len(raw) // 2 + 1:], 'wb'), '-', '0')
    # many records in small segments, a follower that ends early, a follower whose records are longer than the leader's
    many = _write(tmp_path / 'many.fq', ''.join('@r%d/1_RG:Z:g%d\n%s\n+\n%s\n' % (i, i % 5, 'ACGT' * (5 + i // 3000), 'IIII' * (5 + i // 3000)) for i in range(20000)))
    longer = _write(tmp_path / 'longer.fq', ''.join('@r%d/1_RG:Z:g%d and a long comment %s\n%s\n+\n%s\n' % (i, i % 5, 'x' * 200, 'ACGT' * (5 + i // 3000), 'IIII' * (5 + i // 3000)) for i in range(20000)))
    assert 'stream records=20000 usable=20000' in harness('stream', many, longer, '1', '65536')
    short = _write(tmp_path / 'short.fq', ''.join('@r%d/1_RG:Z:g%d\n%s\n+\n%s\n' % (i, i % 5, 'ACGT' * (5 + i // 3000), 'IIII' * (5 + i // 3000)) for i in range(7777)))
    out = harness('stream', many, short, '1', '65536')
    assert 'stream records=20000' in out or 'usable=7777' in out


def test_gammaln_pool(harness):
    out = harness('combiln')
    assert out.count('rc=0') == 10 and 'logtab short rc=0' not in out
    assert 'logtab rc=0 restated differ=0 of' in out         # the restated log over the mapped libm's constants: bit for bit


def test_sam_reader(harness, oracle, tmp_path):
    import oracle_bqsr as OQ
    paths = OQ.synth_bqsr_set(str(tmp_path), seed=3, npairs=400, S=60)
    out = harness('sam', paths['sam'])
    assert 'sam n=800' in out and out.count('rc=0') == 3
    hdr = '@HD\tVN:1.6\n@SQ\tSN:c\tLN:100\n@RG\tID:a\tPU:u\n@RG\tPU:noid\n@CO\tfree text\n'
    ok = 'r1\t99\tc\t5\t60\t4M\t=\t20\t30\tACGT\tIIII\tRG:Z:a\tOQ:Z:JJJJ\n'
    cases = {
        'empty': '', 'header_only': hdr, 'good': hdr + ok, 'no_newline': hdr + ok.rstrip('\n'), 'crlf': (hdr + ok).replace('\n', '\r\n'),
        'few_fields': hdr + 'r1\t99\tc\t5\n', 'bad_flag': hdr + ok.replace('\t99\t', '\tx9\t'), 'bad_pos': hdr + ok.replace('\t5\t60', '\t-\t60'),
        'huge_pos': hdr + ok.replace('\t5\t60', '\t99999999999999999999\t60'), 'star_fields': hdr + 'r1\t4\t*\t0\t0\t*\t*\t0\t0\t*\t*\n',
        'bad_cigar': hdr + ok.replace('4M', '4'), 'cigar_no_len': hdr + ok.replace('4M', 'M'), 'odd_op': hdr + ok.replace('4M', '2M2Z'),
        'huge_cigar_len': hdr + ok.replace('4M', '99999999999M'), 'many_ops': hdr + ok.replace('4M', '1M1I' * 5000),
        'empty_tags': hdr + 'r1\t0\tc\t1\t0\t1M\t*\t0\t0\tA\tI\tRG:Z:\tOQ:Z:\tXX\t\t\n', 'unknown_rg': hdr + ok.replace('RG:Z:a', 'RG:Z:zz'),
        'blank_lines': hdr + '\n\n' + ok + '\n   \n', 'binary': hdr + 'r\x00\t1\tc\x01\t1\t0\t1M\t*\t0\t0\t\xff\t\xfe\n',
        'long_seq': hdr + 'r1\t0\tc\t1\t0\t70000M\t*\t0\t0\t' + 'A' * 70000 + '\t' + 'I' * 70000 + '\n',
        'bam_magic': 'BAM\x01rest', 'tabs_only': '\t\t\t\t\t\t\t\t\t\t\t\n',
    }
    for name, text in cases.items():
        harness('sam', _write(tmp_path / (name + '.sam'), text.encode('latin-1'), 'wb'))
    harness('sam', str(tmp_path / 'missing.sam'))
    # BAM / BGZF / gzip images of the same alignments, intact and damaged (csrc/bam_host.cpp)
    import gzip
    import random
    import bamwriter
    text = open(paths['sam']).read()
    raw = bamwriter.sam_to_bam_bytes(text)
    z = bamwriter.bgzf(raw)
    assert 'sam n=800' in harness('sam', _write(tmp_path / 'ok.bam', z, 'wb'))
    assert 'sam n=800' in harness('sam', _write(tmp_path / 'raw.bam', raw, 'wb'))
    assert 'sam n=800' in harness('sam', _write(tmp_path / 'ok.sam.gz', gzip.compress(text.encode()), 'wb'))
    rng = random.Random(7)
    damaged = {'cut_z': z[:len(z) // 3], 'cut_raw': raw[:len(raw) // 3], 'gz_cut': gzip.compress(text.encode())[:5000],
               'no_records': bamwriter.bgzf(raw[:200]), 'bad_block_size': z[:16] + b'\xff\xff' + z[18:],
               'tiny': b'BAM\1', 'gz_magic_only': b'\x1f\x8b', 'huge_l_text': b'BAM\1\xff\xff\xff\x7f' + b'x' * 64,
               'neg_refs': b'BAM\1\0\0\0\0\xff\xff\xff\xff', 'huge_seq': bamwriter.bgzf(raw[:len(raw) - 900] + b'\x7f' * 900)}
    for k in range(12):                                          # random byte flips in the uncompressed image, re-compressed
        b = bytearray(raw)
        for _ in range(rng.randint(1, 6)):
            b[rng.randrange(len(b))] = rng.randrange(256)
        damaged['flip%d' % k] = bamwriter.bgzf(bytes(b))
    for name, data in damaged.items():
        harness('sam', _write(tmp_path / (name + '.bam'), data, 'wb'))


def test_threaded_readers_under_thread_sanitizer(tmp_path, oracle):
    """The same harness under ThreadSanitizer: the line index, the chunk-parallel scan, the background pair job, the
    fill / format threads, the parked gammaln pool and the SAM reader, on inputs large enough to start every thread."""
    exe = str(tmp_path / 'host_tsan')
    csrc = os.path.join(ROOT, 'kbbq-py_amd', 'csrc')
    cmd = ['g++', '-std=c++17', '-O1', '-g', '-fsanitize=thread', '-pthread', '-o', exe,
           os.path.join(ROOT, 'tests', 'native', 'host_sanitize.cpp'),
           os.path.join(csrc, 'fastq_host.cpp'), os.path.join(csrc, 'fastq_stream.cpp'), os.path.join(csrc, 'solve_host.cpp'),
           os.path.join(csrc, 'sam_host.cpp'), os.path.join(csrc, 'bam_host.cpp'), os.path.join(csrc, 'parallel_gunzip.cpp'), '-lz', '-ldl']
    if subprocess.run(cmd, capture_output=True).returncode != 0:
        pytest.skip('this g++ cannot link -fsanitize=thread')
    n = 20000
    seq, cseq, qual, meta = oracle.synth(0, n, n, 4, 36, 151, 3)
    names = oracle.synth_names(0, n, 3, with_rg=True)
    fa, fb = str(tmp_path / 'a.fq'), str(tmp_path / 'b.fq')
    oracle.write_fastq(fa, names, seq, qual, meta)
    oracle.write_fastq(fb, names, cseq, qual, meta)
    import oracle_bqsr as OQ
    sam = OQ.synth_bqsr_set(str(tmp_path), seed=3, npairs=3000, S=60)['sam']
    import bamwriter
    bam = bamwriter.write_bam(tmp_path / 't.bam', open(sam).read())
    env = dict(os.environ, KBBQ_HOST_THREADS='6', KBBQ_SCAN_CHUNK='501', TSAN_OPTIONS='halt_on_error=0')
    for args in (['pair', fa, fb, '1'], ['pair', fa, '-', '0'], ['stream', fa, fb, '1', '2000000'], ['combiln'], ['sam', sam], ['sam', bam]):
        r = subprocess.run([exe] + args, capture_output=True, text=True, timeout=300, env=env)
        assert 'ThreadSanitizer' not in r.stderr, r.stderr[-3000:]
        assert r.returncode == 0, (args, r.returncode, r.stderr[-2000:])
    assert 'job rc=0 n=%d' % n in subprocess.run([exe, 'pair', fa, fb, '1'], capture_output=True, text=True, env=env).stdout
