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
    return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)


def split_cases(npz):
    """fixtures saved as '<case>/<field>' -> {case: {field: array}}"""
    out = {}
    for k in npz.files:
        if '/' in k:
            c, f = k.split('/', 1)
            out.setdefault(c, {})[f] = npz[k]
    return out


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _verify_spike_tags():
    """Every tensor the model code tags as 'spikes / small integers' is checked for bf16 exactness before a convolution
    uses the one-term path (eas_snn_amd.ops.VERIFY_SMALL_INT)."""
    from eas_snn_amd import ops
    ops.VERIFY_SMALL_INT = True
    yield
    ops.VERIFY_SMALL_INT = False
