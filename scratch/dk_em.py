"""EM divergence probe: GPU vs oracle posteriors on the oracle's WPE output, per EM iteration count."""
import sys
sys.path.insert(0, 'oracle'); sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import gss_oracle as oracle
from pb_chime5_amd import synthetic, ops
from pb_chime5_amd._capi import default_context
import test_gpu_pipeline as tp
D, K = int(sys.argv[1]), int(sys.argv[2])
ctx = default_context(0)
u = synthetic.tiny(seed=D + K, num_channels=D, num_samples=64000, num_speakers=K - 1, context=4096, noise=3e-2)
got, det, want, wdet = tp._run_both(u, wpe=True, wpe_taps=2, wpe_delay=2, wpe_iterations=2, bss_iterations=6)
Obs, act = wdet['Obs'], wdet['activity_freq']
print('Obs', Obs.shape, 'act', act.shape, act.sum(axis=-1))
for it in (1, 2, 3, 4, 6):
    for post in (0, 1):
        g = ops.cacgmm_posteriors(Obs, act, iterations=it, iterations_post=post, ctx=ctx)
        o = oracle.gss_block(Obs, act, iterations=it, iterations_post=post)
        d = np.abs(g - o)
        print(f'it={it} post={post} max {d.max():.3e} frac>1e-6 {(d > 1e-6).mean():.4f}')
