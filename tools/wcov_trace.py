"""Where the M-step kernel (wcov_kernel, D = 24) spends its time: per workgroup (wave 0) shader
cycles in staging / accumulation / reduction and its life span on the 100 MHz clock, from a
build with -DGSS_WCOV_TRACE=1 (tools/build_variant.sh wcovtrace -DGSS_WCOV_TRACE=1).
    python tools/wcov_trace.py [channels=24] [seconds=15]"""
import ctypes
import os
import sys
from pathlib import Path

import numpy as np

R = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(R))
os.environ.setdefault('GSS_HIP_LIBRARY', str(R / 'pb_chime5_amd/lib/variants/libgss_wcovtrace.so'))
from pb_chime5_amd import ops, synthetic               # noqa: E402
from pb_chime5_amd._capi import default_context       # noqa: E402

D = int(sys.argv[1]) if len(sys.argv) > 1 else 24
sec = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
n = int(sec * 16000)
c = n // 3
iv = [(c, n - c), (n // 10, n // 2), (n // 3, n - n // 8), (n // 20, n // 4)]
u = synthetic.make_utterance(11, D, n, iv, start_context=c, end_context=c, rir_taps=1024, noise=3e-2, fast=True)
ctx = default_context(0)
lib = ctypes.CDLL(os.environ['GSS_HIP_LIBRARY'])
lib.gss_debug_wcov_phase.argtypes = [ctypes.c_void_p, ctypes.c_int]
# the EM stage alone, so that the last wcov launch is an M-step (in the pipeline it is the PSD
# launch of the beamformer)
T = n // 256 + 4
rng = np.random.default_rng(0)
Obs = rng.standard_normal((D, T, 513)) + 1j * rng.standard_normal((D, T, 513))
act = np.ones((5, T), dtype=bool)
act[1, : T // 3] = act[2, T // 2:] = act[3, T // 4: T // 2] = False
ops.cacgmm_posteriors(Obs, act, iterations=3, iterations_post=1, ctx=ctx)
ctx.synchronize()


class res:
    pass


res.T = T
buf = np.zeros((8192, 6), dtype=np.int64)
assert lib.gss_debug_wcov_phase(buf.ctypes.data_as(ctypes.c_void_p), 8192) == 0
b = buf[buf[:, 5] > 0].astype(float)
t0 = b[:, 0].min()
start, end = (b[:, 0] - t0) / 100.0, (b[:, 1] - t0) / 100.0      # microseconds
print(f'{len(b)} workgroups of the last launch that wrote (M-step, D = {D}, T = {res.T})')
print(f'  launch span {end.max():.1f} us; workgroup life mean {np.mean(end - start):.1f} us '
      f'(min {np.min(end - start):.1f}, max {np.max(end - start):.1f})')
tot = b[:, 5].mean()
for i, nme in ((2, 'staging (loads, LDS stores, barriers)'), (3, 'accumulation'), (4, 'reduction + store')):
    print(f'  {nme:40s} {b[:, i].mean():10.0f} cycles  {b[:, i].mean() / tot:6.3f}')
print(f'  whole workgroup {tot:.0f} shader cycles')
edges = np.linspace(0, end.max(), 21)
res_t = [(np.sum((start <= t) & (end > t))) for t in edges]
print('  resident workgroups at 20 points of the span:', ' '.join(str(int(x)) for x in res_t))
