"""Where em_chol_kernel (the class update of the D = 24 EM loop, one wave per class matrix) spends
its time: shader-clock stamps per phase from a build with -DGSS_CHOL_TRACE=1
(tools/build_variant.sh choltrace -DGSS_CHOL_TRACE=1).
    python tools/chol_trace.py [channels=24] [seconds=15]"""
import ctypes
import os
import sys
from pathlib import Path

import numpy as np

R = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(R))
os.environ.setdefault('GSS_HIP_LIBRARY', str(R / 'pb_chime5_amd/lib/variants/libgss_choltrace.so'))
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
lib.gss_debug_chol_phase.argtypes = [ctypes.c_void_p, ctypes.c_int]
params = ops.make_params(bss_iterations=20)
ops._prepare_windows(ctx, 1024, 256)
res = ops.ResidentUtterance(ctx, u.obs, u.activity_array, params)
res.enqueue(0, c, c)
ctx.synchronize()
buf = np.zeros((4096, 10), dtype=np.int64)
assert lib.gss_debug_chol_phase(buf.ctypes.data_as(ctypes.c_void_p), 4096) == 0
b = buf[(buf[:, 9] > 0) & (buf[:, 8] > buf[:, 0])].astype(float)
names = ['sum_gamma (one load + wave sum)', 'tri_slots', 'reduce_covariance (chunk partials)',
         'Frobenius norm of B', 'scatter to LDS + register load', 'Cholesky + inverse sweep',
         'ln det', 'B^-1 = W^H W + store', 'flag store']
tot = (b[:, 9] - b[:, 0]).mean()
print(f'{len(b)} waves (last launch), D = {D}, T = {res.T}; mean shader cycles per phase (share):')
for i, nme in enumerate(names):
    d = (b[:, i + 1] - b[:, i]).mean()
    print(f'  {nme:36s} {d:10.0f}  {d / tot:6.3f}')
print(f'  whole wave {tot:.0f} cycles; first start to last end of the launch: '
      f'{b[:, 9].max() - b[:, 0].min():.0f} cycles')
