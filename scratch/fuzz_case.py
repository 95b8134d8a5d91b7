import sys
sys.path.insert(0, 'oracle'); sys.path.insert(0, '.')
import numpy as np
import gss_oracle as oracle
from pb_chime5_amd import ops, synthetic
# seed 7, case 32 of scratch/fuzz.py
u = synthetic.tiny(seed=1032, num_channels=8, num_samples=18481, num_speakers=1, context=1771, noise=5e-2)
kw = dict(wpe=True, wpe_taps=1, wpe_delay=1, wpe_iterations=1, bss_iterations=3, bss_iterations_post=0,
          bf='mvdrSouden_ban', postfilter=None)
got, det = ops.enhance_observation(u.obs, u.activity_array, u.target_index, 1771, 1771, debug=True, **kw)
want, wdet = oracle.enhance_observation(u.obs, u.activity_array, u.target_index, u.ex, return_details=True,
                                        gss_fn=oracle.gss_block_batched, **kw)
cond = np.linalg.cond(wdet['cov_n'])
print('cond(cov_n): median %.1e  max %.1e  frac<1e8 %.2f' % (np.median(cond), cond.max(), (cond < 1e8).mean()))
print('posterior max diff', np.abs(det['posterior'] - wdet['masks']).max() if det['posterior'].shape == wdet['masks'].shape else 'masks zeroed')
err = np.linalg.norm(np.abs(det['X_hat']) - np.abs(wdet['X_hat']), axis=0) / np.maximum(np.linalg.norm(np.abs(wdet['X_hat']), axis=0), 1e-300)
for lo, hi in [(0, 1e6), (1e6, 1e8), (1e8, 1e10), (1e10, 1e30)]:
    sel = (cond >= lo) & (cond < hi)
    if sel.any(): print(f'cond [{lo:.0e},{hi:.0e}): {sel.sum():3d} bins  max err {err[sel].max():.1e}')
