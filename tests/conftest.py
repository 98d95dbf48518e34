"""pytest configuration: `gpu` marker, import paths, shared fixtures."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden')
# the product package (drop-in `kbbq`) lives in kbbq-py_amd/; the oracle in oracle/
for p in (os.path.join(ROOT, 'kbbq-py_amd'), os.path.join(ROOT, 'oracle'), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


GOLDEN_CASES = ['c1_10k_1rg', 'c3cut_2k_8rg', 'c5cut_2k_mixed', 'q42_500_3rg', 'short_64_1rg']


def load_golden(name):
    with open(os.path.join(GOLD, name + '.json')) as fh:
        info = json.load(fh)
    arrs = dict(np.load(os.path.join(GOLD, name + '.npz')))
    return info, arrs


@pytest.fixture(scope='session')
def oracle():
    import oracle as O
    O.lib()
    return O
