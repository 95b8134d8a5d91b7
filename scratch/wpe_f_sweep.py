"""wpe kernel times vs the number of frequencies (tail / round-packing probe), D = 24."""
import sys
sys.path.insert(0, '.')
import numpy as np
from ctypes import c_void_p
from pb_chime5_amd._capi import Context
ctx = Context(0)
D = int(sys.argv[1]) if len(sys.argv) > 1 else 24
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1056
rng = np.random.default_rng(0)
for F in (448, 496, 512, 513, 520, 544, 576, 640):
    Y = (rng.standard_normal((F, T, D)) + 1j * rng.standard_normal((F, T, D)))
    Y_d = ctx.to_device(Y); X_d = ctx.empty(16 * F * T * D)
    def run():
        ctx._check(ctx.lib.gss_wpe(ctx.handle, c_void_p(Y_d.ptr), F, T, D, 10, 2, 3, c_void_p(X_d.ptr)), 'wpe')
    run(); ctx.synchronize()
    ctx.profile_enable(True); ctx.profile_reset()
    for _ in range(3): run()
    prof = ctx.profile_report(); ctx.profile_enable(False)
    line = ' '.join(f"{k[4:]}={v['ms']/v['calls']*1e3/F:.2f}" for k, v in sorted(prof.items()) if k in ('wpe_corr', 'wpe_apply', 'wpe_chol_update', 'wpe_backsolve'))
    print(f'F={F:4d} us/frequency: {line}   corr_ms={prof["wpe_corr"]["ms"]/prof["wpe_corr"]["calls"]:.3f}')
