"""Workgroup timeline of wpe_corr (the LDS-DMA kernel) on the headline shape, from a build with
-DGSS_CORR_TRACE=1 (tools/build_variant.sh corrtrace -DGSS_CORR_TRACE=1): every workgroup
stamps start / first window landed / frame loop done / tile stored on the 100 MHz wall clock.
Answers where the launch's time goes that is not MFMA issue: prologue, epilogue, dispatch
gaps, the tail of the launch.

    GSS_HIP_LIBRARY=pb_chime5_amd/lib/variants/libgss_corrtrace.so python tools/corr_trace.py
"""
import ctypes
import os
import sys
from pathlib import Path

import numpy as np

R = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(R))
os.environ.setdefault('GSS_HIP_LIBRARY', str(R / 'pb_chime5_amd/lib/variants/libgss_corrtrace.so'))
from pb_chime5_amd import ops                      # noqa: E402
from pb_chime5_amd._capi import default_context   # noqa: E402


def main():
    F, T, D = 513, int(os.environ.get('T', 941)), int(os.environ.get('D', 24))
    rng = np.random.default_rng(0)
    ctx = default_context()
    lib = ctypes.CDLL(os.environ['GSS_HIP_LIBRARY'])
    lib.gss_debug_corr_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
    Y = (rng.standard_normal((D, T, F)) + 1j * rng.standard_normal((D, T, F)))
    for _ in range(3):
        ops.wpe_dtf(Y, 10, 2, 1, ctx=ctx)
    ctx.synchronize()
    n = 8192
    buf = np.zeros((n, 6), dtype=np.int64)
    assert lib.gss_debug_corr_trace(buf.ctypes.data_as(ctypes.c_void_p), n) == 0
    used = buf[:, 1] > 0
    tr = buf[used]
    grp, t1, t2, t3, t4 = tr[:, 0], tr[:, 1], tr[:, 2], tr[:, 3], tr[:, 4]
    t0 = t1.min()
    us = lambda x: (x - t0) / 100.0                   # noqa: E731
    span = us(t4.max())
    print(f'F={F} T={T} D={D}: {len(tr)} workgroups, launch span {span:.1f} us')
    dur = (t4 - t1) / 100.0
    pro = (t2 - t1) / 100.0
    loop = (t3 - t2) / 100.0
    epi = (t4 - t3) / 100.0
    print('per group id (tiles sorted heaviest first): count, mean duration us (prologue / loop / store)')
    for g in np.unique(grp):
        m = grp == g
        print(f'  grp {g:2d}: {m.sum():4d}  {dur[m].mean():7.1f}  ({pro[m].mean():5.2f} / {loop[m].mean():7.1f} / {epi[m].mean():5.2f})'
              f'   loop min {loop[m].min():.1f} max {loop[m].max():.1f}')
    # concurrency over time
    ev = np.concatenate([np.stack([us(t1), np.ones(len(tr))], 1), np.stack([us(t4), -np.ones(len(tr))], 1)])
    ev = ev[np.argsort(ev[:, 0], kind='stable')]
    active = np.cumsum(ev[:, 1])
    times = ev[:, 0]
    peak = active.max()
    area = np.sum(active[:-1] * np.diff(times))
    print(f'peak resident workgroups {peak:.0f} ({peak / 256:.2f} per CU); mean resident {area / span:.1f} '
          f'= {area / span / peak:.3f} of peak')
    # when does the launch start to drain?
    full_until = times[np.where(active >= 0.98 * peak)[0][-1]]
    ramp_until = times[np.where(active >= 0.98 * peak)[0][0]]
    print(f'ramp-up to 98 % of peak: {ramp_until:.1f} us; drains from {full_until:.1f} us on '
          f'(tail {span - full_until:.1f} us = {100 * (span - full_until) / span:.1f} % of the span)')
    busy = loop.sum()
    print(f'sum of frame-loop time {busy:.0f} us = {busy / (span * peak):.3f} of span x peak slots; '
          f'prologues {pro.sum() / (span * peak):.3f}, stores {epi.sum() / (span * peak):.3f}')
    # dispatch gaps: per slot we cannot see the slot id, but the gap between one workgroup's
    # end and the next start in time order approximates it
    starts = np.sort(us(t1))
    ends = np.sort(us(t4))
    k = int(peak)
    if len(starts) > k:
        gaps = starts[k:] - ends[:len(starts) - k]
        print(f'start of workgroup i+{k} minus end of the i-th to finish: median {np.median(gaps):.2f} us, '
              f'mean {gaps.mean():.2f} us')
    for q in (0, 5, 50, 95, 100):
        print(f'  duration percentile {q:3d}: {np.percentile(dur, q):7.1f} us')


if __name__ == '__main__':
    main()
