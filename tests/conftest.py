import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + '.npz'))
    return load


@pytest.fixture(scope='session')
def orc():
    """the parity oracle (oracle/fib_oracle.c) — checker only"""
    import oracle
    oracle.build()
    oracle.lib()
    return oracle


@pytest.fixture(scope='session')
def gpu_lib():
    """libfibhip.so loaded + a HIP device present; fails (not skips) if the extension is missing"""
    from fib_tf_amd import _lib
    L = _lib.lib()
    assert L.fibhip_device_count() > 0, 'no HIP device visible: -m gpu tests need an MI355X'
    return _lib
