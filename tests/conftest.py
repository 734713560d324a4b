import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'))


def rel_err(a, b):
    """The project's parity metric: max|a-b| / max|b| over the whole
    array handed in (callers pass one channel, or one frame for PSDs)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    denom = np.max(np.abs(b))
    if denom == 0:
        return float(np.max(np.abs(a)))
    return float(np.max(np.abs(a - b))/denom)


@pytest.fixture(scope='session')
def oracle():
    from oracle import oracle as orc
    orc.build()
    return orc
