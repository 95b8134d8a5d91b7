"""Where the one-launch EM kernel of the one-array regime (em_onchip4_kernel) spends its time:
shader cycles of wave 0 per phase, from a build with -DGSS_EM4_TRACE=1
(tools/build_variant.sh em4trace -DGSS_EM4_TRACE=1).
    python tools/em4_trace.py [seconds=34.7] [iterations=20]"""
import ctypes
import os
import sys
from pathlib import Path

import numpy as np

R = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(R))
os.environ.setdefault('GSS_HIP_LIBRARY', str(R / 'pb_chime5_amd/lib/variants/libgss_em4trace.so'))
from pb_chime5_amd import ops, synthetic               # noqa: E402
from pb_chime5_amd._capi import default_context       # noqa: E402

sec = float(sys.argv[1]) if len(sys.argv) > 1 else 34.7
it = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n = int(sec * 16000)
c = n // 3
iv = [(c, n - c), (n // 10, n // 2), (n // 3, n - n // 8), (n // 20, n // 4)]
u = synthetic.make_utterance(11, 4, n, iv, start_context=c, end_context=c, rir_taps=1024, noise=3e-2, fast=True)
ctx = default_context(0)
lib = ctypes.CDLL(os.environ['GSS_HIP_LIBRARY'])
lib.gss_debug_em4_phase.argtypes = [ctypes.c_void_p, ctypes.c_int]
params = ops.make_params(bss_iterations=it)
ops._prepare_windows(ctx, 1024, 256)
res = ops.ResidentUtterance(ctx, u.obs, u.activity_array, params)
res.enqueue(0, c, c)
ctx.synchronize()
buf = np.zeros((1024, 6), dtype=np.int64)
assert lib.gss_debug_em4_phase(buf.ctypes.data_as(ctypes.c_void_p), 1024) == 0
b = buf[buf[:, 5] > 0].astype(float)
names = ['phase E', 'class update (lanes of wave 0; part of sums + model update)', 'LDS stores', 'phase M', 'sums + model update', 'whole kernel']
print(f'{len(b)} workgroups, T = {res.T}; mean shader cycles of wave 0 (share of the kernel):')
for i, nme in enumerate(names):
    print(f'  {nme:64s} {b[:, i].mean():12.0f}  {b[:, i].mean() / b[:, 5].mean():6.3f}')
print(f'  kernel cycles min / max over workgroups: {b[:, 5].min():.0f} / {b[:, 5].max():.0f}')
w = buf[:, 5].astype(float)
idx = np.nonzero(w > 0)[0]
q = np.percentile(w[idx], [0, 5, 25, 50, 75, 95, 100])
print('  whole-kernel cycles percentiles 0/5/25/50/75/95/100:', ' '.join(f'{x:.0f}' for x in q))
slow = idx[np.argsort(-w[idx])[:24]]
print('  slowest workgroups (blockIdx: cycles):', ' '.join(f'{i}:{w[i]:.0f}' for i in slow))
fast = idx[np.argsort(w[idx])[:12]]
print('  fastest workgroups:', ' '.join(f'{i}:{w[i]:.0f}' for i in fast))
for lo in range(0, 520, 64):
    sel = idx[(idx >= lo) & (idx < lo + 64)]
    print(f'  blockIdx {lo:3d}-{lo + 63:3d}: mean {w[sel].mean():.0f}  E {buf[sel, 0].mean():.0f}  M {buf[sel, 3].mean():.0f}  model {buf[sel, 4].mean():.0f}')
