"""Per-bin error of the fused pipeline against the all-bin oracle next to the oracle's own
last-bit sensitivity, for one of the BASELINE scenes (diagnostic behind the per-bin rule of
tests/test_gpu_pipeline.py):  python tools/per_bin_diag.py {2,3,5} out.npz"""
import sys
from pathlib import Path

import numpy as np

R = Path(__file__).resolve().parents[1]
for p in (str(R), str(R / 'oracle'), str(R / 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)

import gss_oracle as oracle  # noqa: E402
from oracle_pool import OraclePool  # noqa: E402
from pb_chime5_amd import ops, synthetic  # noqa: E402

cfg = int(sys.argv[1])
u = {2: synthetic.config2, 3: lambda: synthetic.config3_item(0), 5: synthetic.config5}[cfg]()
kw = dict(bf='gev_ban', bss_iterations=40) if cfg == 5 else dict(bf='mvdrSouden_ban', bss_iterations=20)
cs = u.ex['start_orig']['original']
ce = u.ex['end']['original'] - u.ex['end_orig']['original']
x_hat, det = ops.enhance_observation(u.obs, u.activity_array, u.target_index, cs, ce, debug=True, **kw)
with OraclePool() as pool:
    okw = dict(return_details=True, gss_fn=pool.gss_block, wpe_fn=pool.wpe_block, **kw)
    want, wdet = oracle.enhance_observation(u.obs, u.activity_array, u.target_index, u.ex, **okw)
    rng = np.random.default_rng(0)
    obs2 = u.obs * (1 + 2e-16 * rng.standard_normal(u.obs.shape))
    _, wdet2 = oracle.enhance_observation(obs2, u.activity_array, u.target_index, u.ex, **okw)
A, B, B2 = np.abs(det['X_hat']), np.abs(wdet['X_hat']), np.abs(wdet2['X_hat'])
nb = np.linalg.norm(B, axis=0)
err = np.linalg.norm(A - B, axis=0) / nb
self_f = np.linalg.norm(B2 - B, axis=0) / nb
cond = np.linalg.cond(wdet['cov_n'])
loud = nb ** 2 > np.max(nb ** 2) * 1e-8
print('ref channel gpu/oracle/oracle2:', det['ref_channel'], wdet['ref_channel'], wdet2['ref_channel'])
print('global |X| err %.2e; bins above 1e-4: %d; above max(1e-4, 10 self): %d; loud %d'
      % (np.max(np.abs(A - B)) / B.max(), (err > 1e-4).sum(), (err > np.maximum(1e-4, 10 * self_f)).sum(), loud.sum()))
for thr in (1e6, 1e8, 1e10, 1e12):
    m = cond < thr
    print('cond < %.0e: %d bins, max err %.2e, max self %.2e' % (thr, m.sum(), err[m].max() if m.any() else 0, self_f[m].max() if m.any() else 0))
worst = np.argsort(-err / np.maximum(self_f, 1e-5))[:12]
for f in worst:
    print('bin %3d err %.2e self %.2e cond %.2e' % (f, err[f], self_f[f], cond[f]))
np.savez(sys.argv[2], err=err, self_f=self_f, cond=cond, nb=nb)
