import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + '.npz')))


def unpack_bits(packed, shape):
    n = int(np.prod(shape))
    return np.unpackbits(packed)[:n].reshape([int(s) for s in shape]).astype(bool)


@pytest.fixture(scope='session')
def head_state():
    from mv2d_amd import synthetic
    return synthetic.make_head_state(seed=0)
