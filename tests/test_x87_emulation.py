"""
CPU test of the 80-bit add emulation used by the device solve (K3): the shared header
kbbq-py_amd/csrc/x87add.h is compiled for the host and compared with the CPU's native x87
long double arithmetic (the arithmetic np.longdouble uses on x86-64) on millions of pairs,
and against NumPy itself on a sample.
"""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT


@pytest.mark.skipif(np.finfo(np.longdouble).nmant != 63, reason='np.longdouble is not x87 extended here')
def test_x87_add_matches_native_long_double(tmp_path):
    exe = str(tmp_path / 'x87_check')
    src = os.path.join(ROOT, 'tests', 'native', 'x87_check.cpp')
    subprocess.check_call(['g++', '-O1', '-std=c++17', '-ffp-contract=off', '-fsanitize=undefined', '-fno-sanitize-recover=undefined', '-o', exe, src])
    out = subprocess.run([exe, '1500000'], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.startswith('OK')
