"""Test infrastructure: the CPU oracle's two per-frequency stages (WPE, CACGMM) spread
over worker processes, so that all-bin oracle runs of the large BASELINE configs finish
in a minute instead of ten.  Frequencies are independent in both stages (the reference
loops over them, /root/reference/pb_chime5/core.py:172), so splitting the frequency axis
changes nothing in the arithmetic."""
import multiprocessing as mp
import os
import sys
from concurrent.futures import ProcessPoolExecutor
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent


def _init():
    # one BLAS thread per worker, like the reference's own processes
    # (pb_chime5/__init__.py:3-14).  NumPy is already imported when this runs (unpickling
    # this function imports the module), so the limit is set at run time, not through the
    # environment -- which would leak into everything the test process starts later.
    import threadpoolctl
    global _LIMIT
    _LIMIT = threadpoolctl.threadpool_limits(1)
    for p in (str(REPO), str(REPO / 'oracle')):
        if p not in sys.path:
            sys.path.insert(0, p)


def _wpe(job):
    import gss_oracle as oracle
    Obs, taps, delay, iterations, psd_context = job
    return oracle.wpe_block(Obs, taps, delay, iterations, psd_context)


def _gss(job):
    import gss_oracle as oracle
    Obs, act, iterations, iterations_post = job
    return oracle.gss_block_batched(Obs, act, iterations=iterations,
                                    iterations_post=iterations_post)


def usable_cpus():
    n = len(os.sched_getaffinity(0))
    try:
        q, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if q != 'max':
            n = min(n, max(1, int(float(q) / float(period))))
    except (OSError, ValueError):
        pass
    return n


class OraclePool:
    """``wpe_fn`` / ``gss_fn`` drop-ins for oracle.enhance_observation."""

    def __init__(self, workers=None, bins_per_job=12):
        self.workers = workers or max(1, min(14, usable_cpus() - 2))
        self.bins_per_job = bins_per_job
        self._ex = ProcessPoolExecutor(self.workers, mp_context=mp.get_context('spawn'),
                                       initializer=_init)

    def close(self):
        self._ex.shutdown(wait=True, cancel_futures=True)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _map(self, fn, Obs, make_job):
        F = Obs.shape[-1]
        edges = list(range(0, F, self.bins_per_job)) + [F]
        futures = [self._ex.submit(fn, make_job(np.ascontiguousarray(Obs[..., a:b])))
                   for a, b in zip(edges[:-1], edges[1:])]
        return np.concatenate([f.result(timeout=3600) for f in futures], axis=-1)

    def wpe_block(self, Obs, taps=10, delay=2, iterations=3, psd_context=0):
        return self._map(_wpe, Obs, lambda o: (o, taps, delay, iterations, psd_context))

    def gss_block(self, Obs, activity_freq, iterations=20, iterations_post=1):
        act = np.asarray(activity_freq)
        return self._map(_gss, Obs, lambda o: (o, act, iterations, iterations_post))
