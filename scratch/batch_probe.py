import sys, time, ctypes
sys.path.insert(0, '/root/repo')
import numpy as np
from pb_chime5_amd._capi import Context, c_void_p
ctx = Context(0)
lib = ctx.lib
T, D, K = 941, 24, 5
rng = np.random.default_rng(0)
def run(F, reps=3):
    Y = (rng.standard_normal((F, T, D)) + 1j * rng.standard_normal((F, T, D)))
    # add some temporal correlation so WPE is well posed
    Y[:, 1:] += 0.5 * Y[:, :-1]
    Y_d = ctx.to_device(Y); X_d = ctx.empty(Y.nbytes)
    act = np.ones((K, T), np.uint8); act[0, :300] = 0; act[1, 500:] = 0
    act_d = ctx.to_device(act); g_d = ctx.empty(8 * F * K * T)
    mx = ctx.empty(8*F*T); mn = ctx.empty(8*F*T); xh = ctx.empty(16*F*T); ref = ctx.empty(16)
    def once():
        ctx._check(lib.gss_wpe(ctx.handle, c_void_p(Y_d.ptr), F, T, D, 10, 2, 3, c_void_p(X_d.ptr)), 'wpe')
        ctx._check(lib.gss_cacgmm(ctx.handle, c_void_p(X_d.ptr), F, T, D, c_void_p(act_d.ptr), K, 20, 1, c_void_p(g_d.ptr)), 'em')
        ctx._check(lib.gss_masks_from_posteriors(ctx.handle, c_void_p(g_d.ptr), F, K, T, 0, 1, 100, 100, c_void_p(mx.ptr), c_void_p(mn.ptr)), 'masks')
        ctx._check(lib.gss_mvdr_souden(ctx.handle, c_void_p(X_d.ptr), F, T, D, c_void_p(mx.ptr), c_void_p(mn.ptr), 1, c_void_p(xh.ptr), c_void_p(ref.ptr)), 'mvdr')
    once(); ctx.synchronize()
    t = time.perf_counter()
    for _ in range(reps): once()
    ctx.synchronize()
    return (time.perf_counter() - t) / reps * 1e3
for F in (513, 1026, 2052):
    ms = run(F)
    print(f'F={F}: {ms:.2f} ms  -> per 513-bin utterance {ms * 513 / F:.2f} ms')
