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


def rel_err_per_freq(a, b, axis=-1, floor_db=-80.0):
    """max over the frequency bins (along ``axis``) of ||a_f - b_f|| / ||b_f||, over the bins
    whose energy ||b_f||^2 lies within ``floor_db`` of the loudest bin.  Unlike ``rel_err``
    (one global scale: the loudest bins decide) every bin is held to the tolerance on its
    own scale."""
    a = np.moveaxis(np.asarray(a), axis, 0)
    b = np.moveaxis(np.asarray(b), axis, 0)
    assert a.shape == b.shape, (a.shape, b.shape)
    nb = np.sqrt(np.sum(np.abs(b.reshape(b.shape[0], -1)) ** 2, axis=1))
    nd = np.sqrt(np.sum(np.abs((a - b).reshape(b.shape[0], -1)) ** 2, axis=1))
    keep = nb ** 2 > np.max(nb ** 2) * 10.0 ** (floor_db / 10.0)
    assert keep.any()
    return float(np.max(nd[keep] / nb[keep]))


@pytest.fixture(scope='session')
def golden():
    def load(name):
        return np.load(GOLDEN / name, allow_pickle=False)
    return load


@pytest.fixture(scope='session')
def oracle_pool():
    """Worker processes for all-bin oracle runs (tests/oracle_pool.py)."""
    sys.path.insert(0, str(REPO / 'tests'))
    from oracle_pool import OraclePool
    pool = OraclePool()
    yield pool
    pool.close()


@pytest.fixture(scope='session')
def gpu_ctx():
    from pb_chime5_amd._capi import default_context
    return default_context(0)


@pytest.fixture(scope='session')
def ref_mismatches():
    """Every case in which the GPU picked another reference channel than the oracle (each one
    certified as a tie or a degenerate scene and re-checked with the oracle's channel forced,
    tests/test_gpu_pipeline.py::_check_ref_channel_or_tie).  Bounded when the session ends --
    independent of test order, -k selections and xdist."""
    found = []
    yield found
    print('reference-channel mismatches:', found)
    # one is known and certified ((D, K) = (24, 2): one point source on 24 microphones,
    # cond(Phi_N) = 3e18, a last-bit move of 5e46 -- DESIGN.md section 4); a second one is news
    assert len(found) <= 1, found
