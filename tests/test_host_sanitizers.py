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
           os.path.join(csrc, 'fastq_host.cpp'), os.path.join(csrc, 'solve_host.cpp')]
    subprocess.check_call(cmd)

    def run(*args):
        env = dict(os.environ, ASAN_OPTIONS='detect_leaks=1:abort_on_error=0', UBSAN_OPTIONS='halt_on_error=1:print_stacktrace=1',
                   KBBQ_HOST_THREADS='4')
        r = subprocess.run([exe] + list(args), capture_output=True, text=True, timeout=300, env=env)
        assert 'ERROR: AddressSanitizer' not in r.stderr and 'runtime error' not in r.stderr, r.stderr[-3000:]
        assert r.returncode == 0, (r.returncode, r.stderr[-2000:])
        return r.stdout
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
    out = harness('pair', fa, '-', '0')
    assert 'scan rc=0' in out


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
    }
    good = _write(tmp_path / 'good.fq', '@r1/1_RG:Z:a\nACGT\n+\nIIII\n@r2/2_RG:Z:a\nACGT\n+\nIIII\n')
    for name, text in cases.items():
        p = _write(tmp_path / (name + '.fq'), text.encode('latin-1'), 'wb')
        for infer in ('0', '1'):
            harness('pair', p, '-', infer)
            harness('pair', p, good, infer)
            harness('pair', good, p, infer)
    harness('pair', str(tmp_path / 'does_not_exist.fq'), '-', '0')


def test_gammaln_pool(harness):
    out = harness('combiln')
    assert out.count('rc=0') == 5
