"""
The file path in constant device memory (kbbq/_stream.py) and over inputs that are read sequentially (fastx.FastqStream):
same bytes, same count tables, same errors at the same reads as the resident path -- which the goldens of the unmodified
reference pin (tests/golden/) -- with the reads going through K1 and K2 slab by slab.
  KBBQ_DEVICE_BUDGET   what a shard may hold on the device (tiny here: hundreds of slabs per golden)
  KBBQ_SEQUENTIAL=1    read regular files the way pipes are read; KBBQ_SEGMENT_BYTES: text per segment
Reference: recalibrate.py:56-57 (zip over two FastxFile walks), :141-156 (the second walk over file A).
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN_CASES, ROOT, load_golden
from test_gpu_parity import VEC, _files, dev          # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


def _capture(fn):
    """Bytes fn() writes to sys.stdout (a binary-backed stand-in, so that the egress pipeline takes its write(2) route)."""
    import io
    import tempfile
    with tempfile.TemporaryFile() as tmp:
        saved = sys.stdout
        sys.stdout = io.TextIOWrapper(io.FileIO(tmp.fileno(), 'wb', closefd=False), write_through=True)
        try:
            fn()
            sys.stdout.flush()
        finally:
            sys.stdout = saved
        tmp.seek(0)
        return tmp.read()


@pytest.mark.parametrize('mode', ['budget', 'sequential'])
@pytest.mark.parametrize('name', GOLDEN_CASES)
def test_streamed_passes_give_the_reference_bytes(dev, oracle, name, mode, tmp_path, monkeypatch):
    """The five goldens with a device budget of 1 MB (slabs of a few thousand reads) and, `sequential`, read segment by
    segment (64 KB of text each) like a pipe: the reference's count vectors and the reference's output bytes; what the run
    held on the device stays within the budget + the count tables + the LUT."""
    from kbbq import recalibrate
    info, gold = load_golden(name)
    fa, fb = _files(oracle, info, tmp_path)
    infer = info['case']['infer_rg']
    monkeypatch.setenv('KBBQ_DEVICE_BUDGET', '1M')
    if mode == 'sequential':
        monkeypatch.setenv('KBBQ_SEQUENTIAL', '1')
        monkeypatch.setenv('KBBQ_SEGMENT_BYTES', '64K')
    vec = recalibrate.fastq_to_covariate_arrays([fa, fb], infer_rg=infer)
    for k, v in zip(VEC, vec):
        assert np.array_equal(v, gold[k]), k
    text = _capture(lambda: recalibrate.recalibrate_fastq([fa, fb], infer_rg=infer))
    assert len(text) == info['output_len'] and oracle.sha256(text) == info['output_sha256']
    st = recalibrate.LAST_RUN.get('streamed')
    if st is None:                                               # the tiny cases fit the budget: they stay resident unless read sequentially
        assert mode == 'budget' and info['case']['n'] <= 500
    else:
        assert st['reads'] == info['case']['n'] and bool(st.get('sequential')) == (mode == 'sequential')
        slack = 3 * st['tables_bytes'] + 2 * st['lut_bytes'] + (1 << 18)      # the tables, a slab's own tables, the snapshot; LUTs; small scratch
        assert st['peak_device_bytes'] <= st['device_budget_bytes'] + slack, st
    out = str(tmp_path / 'out.fq')
    recalibrate.recalibrate_fastq([fa, fb], infer_rg=infer, output=out)
    assert oracle.sha256(open(out, 'rb').read()) == info['output_sha256']


def _cli(args, env=None, timeout=300, shell=False):
    e = dict(os.environ, PYTHONPATH=os.path.join(ROOT, 'kbbq-py_amd'))
    e.update(env or {})
    return subprocess.run(args, env=e, capture_output=True, timeout=timeout, shell=shell, executable='/bin/bash' if shell else None)


@pytest.mark.parametrize('name', ['c1_10k_1rg', 'c3cut_2k_8rg', 'c5cut_2k_mixed'])
def test_the_command_line_reads_pipes(dev, oracle, name, tmp_path):
    """`kbbq recalibrate -f <(cat a.fq) <(cat b.fq)` (two process substitutions: neither input can be mapped, sought or read
    twice -- file A is spooled for pass 2) and `-f - b.fq` with file A on standard input print the golden; gzip-compressed
    input likewise -- decompressed in the pipe, carried by the pipe, or a .gz file read segment by segment; the torch-free command
    line streams within a budget too."""
    import gzip, shutil
    info, _ = load_golden(name)
    fa, fb = _files(oracle, info, tmp_path)
    rg = ' --infer-rg' if info['case']['infer_rg'] else ''
    py = sys.executable
    r = _cli('%s -m kbbq.main recalibrate -f <(cat %s) <(cat %s)%s' % (py, fa, fb, rg), shell=True, env={'KBBQ_SEGMENT_BYTES': '300K'})
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert len(r.stdout) == info['output_len'] and oracle.sha256(r.stdout) == info['output_sha256']
    r = _cli('cat %s | %s -m kbbq.main recalibrate -f - %s%s' % (fa, py, fb, rg), shell=True)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert oracle.sha256(r.stdout) == info['output_sha256']
    with open(fb, 'rb') as src, gzip.open(fb + '.gz', 'wb') as dst:
        shutil.copyfileobj(src, dst)
    out = str(tmp_path / 'o.fq')
    r = _cli('%s -m kbbq.main recalibrate -f %s <(zcat %s.gz)%s -o %s' % (py, fa, fb, rg, out), shell=True)
    assert r.returncode == 0 and not r.stdout, r.stderr.decode()[-2000:]
    assert oracle.sha256(open(out, 'rb').read()) == info['output_sha256']
    # compressed bytes in a pipe, and a compressed FILE read segment by segment (as files of 1 GB and more are): inflated as they are read
    with open(fa, 'rb') as src, gzip.open(fa + '.gz', 'wb') as dst:
        shutil.copyfileobj(src, dst)
    r = _cli('%s -m kbbq.main recalibrate -f %s <(cat %s.gz)%s' % (py, fa, fb, rg), shell=True)
    assert r.returncode == 0 and oracle.sha256(r.stdout) == info['output_sha256'], r.stderr.decode()[-2000:]
    r = _cli([py, '-m', 'kbbq.main', 'recalibrate', '-f', fa + '.gz', fb + '.gz'] + rg.split(), env={'KBBQ_GZ_STREAM_BYTES': '1', 'KBBQ_SEGMENT_BYTES': '200K', 'KBBQ_TIMING': '1'})
    assert r.returncode == 0 and oracle.sha256(r.stdout) == info['output_sha256'], r.stderr.decode()[-2000:]
    assert b'open+index+scan' not in r.stderr and b' scan ' in r.stderr           # no mapped, indexed reader took part
    # regular files over budget, no torch
    r = _cli([py, '-X', 'importtime', '-m', 'kbbq.main', 'recalibrate', '-f', fa, fb] + rg.split(), env={'KBBQ_DEVICE_BUDGET': '2M'})
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    mods = {ln.split('|')[-1].strip() for ln in r.stderr.decode().split('\n') if ln.startswith('import time:')}
    assert 'kbbq._stream' in mods and not any(m == 'torch' for m in mods)
    assert oracle.sha256(r.stdout) == info['output_sha256']


def _pair(oracle, tmp_path, n=6000, lo=60, hi=150, nrg=3, seed=5):
    seq, cseq, qual, meta = oracle.synth(0, n, n, seed, lo, hi, nrg)
    order = np.argsort(meta & 0xFFFF, kind='stable')
    seq, cseq, qual, meta = seq[order], cseq[order], qual[order], meta[order]
    names = oracle.synth_names(0, n, nrg, with_rg=True)
    fa, fb = str(tmp_path / 'a.fq'), str(tmp_path / 'b.fq')
    oracle.write_fastq(fa, names, seq, qual, meta)
    oracle.write_fastq(fb, names, cseq, qual, meta)
    return fa, fb


MODES = {'resident': {}, 'budget': {'KBBQ_DEVICE_BUDGET': '1M'},
         'sequential': {'KBBQ_DEVICE_BUDGET': '1M', 'KBBQ_SEQUENTIAL': '1', 'KBBQ_SEGMENT_BYTES': '100K'}}


def _in_mode(monkeypatch, mode):
    for k in ('KBBQ_DEVICE_BUDGET', 'KBBQ_SEQUENTIAL', 'KBBQ_SEGMENT_BYTES'):
        monkeypatch.delenv(k, raising=False)
    for k, v in MODES[mode].items():
        monkeypatch.setenv(k, v)


def test_streamed_passes_stop_at_the_reads_the_resident_path_stops_at(dev, oracle, tmp_path, monkeypatch):
    """Bad input in the middle of a file -- a quality above 42, a foreign letter, a read shorter than an earlier one, a
    corrected read with another name, a read group that cannot be inferred: the same exception class as the resident
    path (= the reference's), naming the same read, in every mode."""
    from kbbq import recalibrate
    fa, fb = _pair(oracle, tmp_path)
    A = open(fa).read().split('\n')
    B = open(fb).read().split('\n')

    def edit(which, rec, line, fn):
        L = list(A if which == 'a' else B)
        L[4 * rec + line] = fn(L[4 * rec + line])
        p = str(tmp_path / ('bad_%s.fq' % which))
        open(p, 'w').write('\n'.join(L))
        return (p, fb) if which == 'a' else (fa, p)
    cases = {
        'q43': edit('a', 3711, 3, lambda q: q[:7] + 'L' + q[8:]),
        'letter': edit('a', 2503, 1, lambda s_: s_[:5] + 'x' + s_[6:]),
        'name': edit('b', 4100, 0, lambda h: '@zz' + h[3:]),
        'norg': edit('a', 1999, 0, lambda h: h.split('_')[0]),
    }
    short = list(A)
    short[4 * 5000 + 1] = short[4 * 5000 + 1][:40]; short[4 * 5000 + 3] = short[4 * 5000 + 3][:40]
    shortb = list(B)
    shortb[4 * 5000 + 1] = shortb[4 * 5000 + 1][:40]; shortb[4 * 5000 + 3] = shortb[4 * 5000 + 3][:40]
    pa, pb = str(tmp_path / 'short_a.fq'), str(tmp_path / 'short_b.fq')
    open(pa, 'w').write('\n'.join(short)); open(pb, 'w').write('\n'.join(shortb))
    cases['short'] = (pa, pb)
    for what, pair in cases.items():
        seen = {}
        for mode in MODES:
            _in_mode(monkeypatch, mode)
            with pytest.raises((IndexError, TypeError, AssertionError, ValueError)) as ei:
                recalibrate.fastq_to_covariate_arrays(list(pair), infer_rg=True)
            seen[mode] = (type(ei.value), str(ei.value))
            with pytest.raises(type(ei.value)):
                _capture(lambda: recalibrate.recalibrate_fastq(list(pair), infer_rg=True))
        assert seen['budget'][0] is seen['resident'][0] and seen['sequential'][0] is seen['resident'][0], (what, seen)
        # the messages that name a read name the same one
        import re
        nums = {m: re.findall(r'read (\d+)', v[1]) for m, v in seen.items()}
        assert nums['budget'] == nums['resident'] == nums['sequential'], (what, seen)


def test_streamed_pass_2_covers_all_of_file_a(dev, oracle, tmp_path, monkeypatch):
    """File B shorter than file A (zip() stops pass 1, pass 2 prints all of A with its own read-group numbering) and the
    model file (-g: saved by one run, loaded by the next -- file B is then not read at all): the oracle's text in every mode;
    a read group that only appears behind file B's end is the reference's IndexError in every mode."""
    from kbbq import recalibrate
    fa, fb = _pair(oracle, tmp_path, n=5000)
    lines = open(fb).read().split('\n')
    short = str(tmp_path / 'b_short.fq')
    open(short, 'w').write('\n'.join(lines[:4 * 2400]) + '\n')
    want_short = oracle.recalibrate_fastq_text([fa, short], True)[0]
    want = oracle.recalibrate_fastq_text([fa, fb], True)[0]
    for mode in MODES:
        _in_mode(monkeypatch, mode)
        assert _capture(lambda: recalibrate.recalibrate_fastq([fa, short], infer_rg=True)).decode() == want_short, mode
        model = str(tmp_path / ('model_%s.txt' % mode))
        assert _capture(lambda: recalibrate.recalibrate_fastq([fa, fb], infer_rg=True, gatkreport=model)).decode() == want, mode
        assert os.path.getsize(model) > 500
        assert _capture(lambda: recalibrate.recalibrate_fastq([fa, '/nonexistent'], infer_rg=True, gatkreport=model)).decode() == want, mode
    # a new read group behind file B's end
    A = open(fa).read().split('\n')
    A[4 * 4000] = A[4 * 4000].rsplit(':', 1)[0] + ':late'
    A[4 * 4001] = A[4 * 4001].rsplit(':', 1)[0] + ':late'
    late = str(tmp_path / 'late.fq')
    open(late, 'w').write('\n'.join(A))
    for mode in MODES:
        _in_mode(monkeypatch, mode)
        with pytest.raises(IndexError):
            _capture(lambda: recalibrate.recalibrate_fastq([late, short], infer_rg=True))


def test_two_ranks_stream_their_shards(dev, oracle, tmp_path):
    """Two ranks (gloo rehearsal on this GPU), each with a device budget far below its shard: byte ranges cut as ever,
    slabs through K1, one allreduce, slabs through K2: the golden bytes, to stdout in rank order and to per-rank files."""
    import glob
    from test_gpu_pairs import _run_ranks
    info, _ = load_golden('c1_10k_1rg')
    fa, fb = _files(oracle, info, tmp_path)
    r = _run_ranks(2, ['recalibrate', '-f', fa, fb], env={'KBBQ_DEVICE_BUDGET': '1M'})
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert oracle.sha256(r.stdout.decode()) == info['output_sha256']
    out = str(tmp_path / 'o.fq')
    r = _run_ranks(2, ['recalibrate', '-f', fa, fb, '-o', out], env={'KBBQ_DEVICE_BUDGET': '1M'})
    assert r.returncode == 0 and r.stdout == b'', r.stderr.decode()[-2000:]
    text = b''.join(open(p, 'rb').read() for p in sorted(glob.glob(out + '.rank*')))
    assert oracle.sha256(text.decode()) == info['output_sha256']
    # sequentially read inputs cannot be cut into byte ranges: refused, not mis-read
    r = _run_ranks(2, ['recalibrate', '-f', fa, fb], env={'KBBQ_SEQUENTIAL': '1'})
    assert r.returncode != 0 and b'single process' in r.stderr


def test_wrapped_fastq_through_the_command_line(dev, oracle, tmp_path):
    """A FASTQ pair whose sequence and quality lines are wrapped at 70 characters (kseq / pysam.FastxFile reads such files,
    recalibrate.py:56): the command line prints the golden, records four lines each as the reference prints them."""
    from test_host_logic import _wrapped
    info, _ = load_golden('c1_10k_1rg')
    fa, fb = _files(oracle, info, tmp_path)
    wa, wb = str(tmp_path / 'wa.fq'), str(tmp_path / 'wb.fq')
    _wrapped(fa, wa, 70, plus_name=True); _wrapped(fb, wb, 70)
    r = _cli([sys.executable, '-m', 'kbbq.main', 'recalibrate', '-f', wa, wb])
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert len(r.stdout) == info['output_len'] and oracle.sha256(r.stdout) == info['output_sha256']


@pytest.mark.parametrize('mode', ['mapped', 'sequential', 'back_to_zlib'])
@pytest.mark.parametrize('name', ['c1_10k_1rg', 'c3cut_2k_8rg', 'c5cut_2k_mixed'])
def test_gzip_pairs_inflated_on_many_threads_give_the_reference_bytes(dev, oracle, name, mode, tmp_path, monkeypatch):
    """The goldens as `.fq.gz` pairs -- what pysam.FastxFile reads transparently (recalibrate.py:56,141) -- with every gzip member cut
    into 4 KB chunks that inflate side by side (csrc/parallel_gunzip.cpp): through the mapped reader, through the sequential reader
    (file A inflated again in pass 2: no spool), and with the decoder giving up after its second window (zlib takes over from the
    file's start, past what was handed out): the reference's count vectors and output bytes."""
    import gzip
    from kbbq import recalibrate
    info, gold = load_golden(name)
    fa, fb = _files(oracle, info, tmp_path)
    for p in (fa, fb):
        with open(p, 'rb') as src, open(p + '.gz', 'wb') as dst:
            dst.write(gzip.compress(src.read(), 6))
    infer = info['case']['infer_rg']
    monkeypatch.setenv('KBBQ_PGZ_MIN_BYTES', '0')
    monkeypatch.setenv('KBBQ_PGZ_CHUNK', '4096')
    monkeypatch.setenv('KBBQ_HOST_THREADS', '4')
    if mode != 'mapped':
        monkeypatch.setenv('KBBQ_SEQUENTIAL', '1')
        monkeypatch.setenv('KBBQ_SEGMENT_BYTES', '128K')
    if mode == 'back_to_zlib':
        monkeypatch.setenv('KBBQ_PGZ_TEST_FAIL_AFTER', '2')
    vec = recalibrate.fastq_to_covariate_arrays([fa + '.gz', fb + '.gz'], infer_rg=infer)
    for k, v in zip(VEC, vec):
        assert np.array_equal(v, gold[k]), k
    out = str(tmp_path / 'out.fq')
    recalibrate.recalibrate_fastq([fa + '.gz', fb + '.gz'], infer_rg=infer, output=out)
    assert oracle.sha256(open(out, 'rb').read()) == info['output_sha256']
    if mode != 'mapped':
        assert recalibrate.LAST_RUN['streamed']['sequential'] and not recalibrate.LAST_RUN['streamed']['spooled']
