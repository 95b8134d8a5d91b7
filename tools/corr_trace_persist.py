"""Timeline of the persistent wpe_corr kernel (build: tools/build_variant.sh corrtrace
-DGSS_CORR_TRACE=1): per resident workgroup the XCD, start / end on the 100 MHz wall clock and
on the shader clock (s_memtime), and the number of items it took.  Prints the spread of the
finish times (the tail), the items per workgroup, and the SHADER CLOCK the kernel really runs at
inside the full pipeline (the chip lowers it under sustained f64 MFMA load).
    python tools/corr_trace_persist.py"""
import ctypes
import os
import sys
from pathlib import Path

import numpy as np

R = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(R))
os.environ.setdefault('GSS_HIP_LIBRARY', str(R / 'pb_chime5_amd/lib/variants/libgss_corrtrace.so'))
from pb_chime5_amd import ops, synthetic               # noqa: E402
from pb_chime5_amd._capi import default_context       # noqa: E402


def main():
    ctx = default_context()
    lib = ctypes.CDLL(os.environ['GSS_HIP_LIBRARY'])
    lib.gss_debug_corr_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
    utt = synthetic.config2()
    params = ops.make_params()
    ops._prepare_windows(ctx, params.stft_size, params.stft_shift)
    res = ops.ResidentUtterance(ctx, utt.obs, utt.activity_array, params)
    c = utt.ex['start_orig']['original']
    steps = int(os.environ.get('STEPS', 12))
    lib.gss_debug_corr_phase.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    ph = np.zeros((8192, 4), dtype=np.int64)
    res.enqueue(utt.target_index, c, c)
    ctx.synchronize()
    lib.gss_debug_corr_phase(ph.ctypes.data_as(ctypes.c_void_p), 8192, 1)
    for _ in range(steps):                    # sustained load, like the bench
        res.enqueue(utt.target_index, c, c)
    ctx.synchronize()
    lib.gss_debug_corr_phase(ph.ctypes.data_as(ctypes.c_void_p), 8192, 0)
    ph = ph[ph.sum(1) > 0].astype(float)
    tot = ph.sum()
    print('wave 0 of every workgroup, share of the frame loop: issue %.3f, MFMA chunk %.3f, DMA wait %.3f, '
          'barrier %.3f' % tuple(ph.sum(0) / tot))
    n = 8192
    buf = np.zeros((n, 6), dtype=np.int64)
    assert lib.gss_debug_corr_trace(buf.ctypes.data_as(ctypes.c_void_p), n) == 0
    tr = buf[buf[:, 1] > 0]
    xcd, r0, c0, r1, c1, items = tr.T
    us = lambda x: (x - r0.min()) / 100.0               # noqa: E731
    span = us(r1).max()
    print(f'{len(tr)} resident workgroups, span {span:.1f} us, items per workgroup '
          f'{items.min()}..{items.max()} (sum {items.sum()})')
    print('workgroups per XCD:', np.bincount(xcd, minlength=8).tolist())
    end = us(r1)
    print(f'finish times: min {end.min():.1f}  5 % {np.percentile(end, 5):.1f}  median {np.median(end):.1f}  '
          f'max {end.max():.1f} us -> mean idle at the end {np.mean(end.max() - end):.1f} us '
          f'= {100 * np.mean(end.max() - end) / span:.1f} % of the span')
    print(f'start times: max {us(r0).max():.1f} us')
    ghz = (c1 - c0) / ((r1 - r0) * 10.0)
    print(f'shader clock over the workgroups\' lifetimes: mean {ghz.mean():.3f} GHz '
          f'(min {ghz.min():.3f}, max {ghz.max():.3f})')
    for x in range(8):
        m = xcd == x
        if m.any():
            print(f'  XCD {x}: {m.sum():3d} workgroups, {items[m].sum():4d} items, last finish {end[m].max():.1f} us, '
                  f'clock {ghz[m].mean():.3f} GHz')


if __name__ == '__main__':
    main()
