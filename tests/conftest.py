import os
import sys
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parent.parent
GOLDEN = REPO / 'tests' / 'golden'
for p in (str(REPO), str(REPO / 'oracle')):
    if p not in sys.path:
        sys.path.insert(0, p)

# the reference pins its numeric libraries to one thread (pb_chime5/__init__.py:3-14)
for var in ('OMP_NUM_THREADS', 'MKL_NUM_THREADS', 'OPENBLAS_NUM_THREADS'):
    os.environ.setdefault(var, '1')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs an AMD GPU (run with -m gpu)')


def rel_err(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    if a.size == 0:
        return 0.0
    scale = np.max(np.abs(b))
    if scale == 0:
        return float(np.max(np.abs(a)))
    return float(np.max(np.abs(a - b)) / scale)


@pytest.fixture(scope='session')
def golden():
    def load(name):
        return np.load(GOLDEN / name, allow_pickle=False)
    return load


@pytest.fixture(scope='session')
def gpu_ctx():
    from pb_chime5_amd._capi import default_context
    return default_context(0)
