"""The WPE kernels alone (HIP events around every launch) on random data of one shape -- for
A/B runs of wpe_corr / solve / apply variants, also builds whose results are garbage (timing-only
ablations): nothing downstream looks at the output.

    [GSS_HIP_LIBRARY=...] python tools/wpe_kprof.py [D=24] [T=941] [F=513] [reps=6]
"""
import sys
from ctypes import c_void_p
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
from pb_chime5_amd._capi import default_context

D = int(sys.argv[1]) if len(sys.argv) > 1 else 24
T = int(sys.argv[2]) if len(sys.argv) > 2 else 941
F = int(sys.argv[3]) if len(sys.argv) > 3 else 513
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 6
rng = np.random.default_rng(0)
ctx = default_context(0)
Y = (rng.standard_normal((F, T, D)) + 1j * rng.standard_normal((F, T, D)))
Y_d = ctx.to_device(Y)
X_d = ctx.empty(16 * F * T * D)


def run():
    ctx._check(ctx.lib.gss_wpe(ctx.handle, c_void_p(Y_d.ptr), F, T, D, 10, 2, 3, 0, c_void_p(X_d.ptr)), 'gss_wpe')


for _ in range(2):
    run()
ctx.synchronize()
ctx.profile_enable(True)
ctx.profile_reset()
for _ in range(reps):
    run()
prof = ctx.profile_report()
ctx.profile_enable(False)
print(f'D={D} T={T} F={F}: ' + '  '.join(
    f"{k} {v['ms'] / v['calls']:.4f}x{v['calls'] // reps}" for k, v in sorted(prof.items(), key=lambda kv: -kv[1]['ms'])))
