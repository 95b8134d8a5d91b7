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


def main():
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
    Xg = det['X_hat']
    if kw['bf'] == 'mvdrSouden_ban' and det['ref_channel'] != wdet['ref_channel']:
        print('reference channel differs: forcing the oracle\'s on the GPU tensors')
        Xg = ops.mvdr_souden_from_masks(det['Obs'], det['target_mask'], det['distortion_mask'],
                                        ban=True, ref_channel=wdet['ref_channel'])
    A, B, B2 = np.abs(Xg), np.abs(wdet['X_hat']), np.abs(wdet2['X_hat'])
    nb = np.linalg.norm(B, axis=0)
    err = np.linalg.norm(A - B, axis=0) / nb
    self_f = np.linalg.norm(B2 - B, axis=0) / nb
    Yf = wdet['Obs'].transpose(2, 0, 1)
    cond = np.linalg.cond(oracle.get_power_spectral_density_matrix(Yf, wdet['distortion_mask'].T))
    wpe_err = np.linalg.norm(det['Obs'] - wdet['Obs'], axis=(0, 1)) / np.linalg.norm(wdet['Obs'], axis=(0, 1))
    wpe_self = np.linalg.norm(wdet2['Obs'] - wdet['Obs'], axis=(0, 1)) / np.linalg.norm(wdet['Obs'], axis=(0, 1))
    m_err = np.max(np.abs(det['target_mask'] - wdet['target_mask']), axis=0)
    m_self = np.max(np.abs(wdet2['target_mask'] - wdet['target_mask']), axis=0)
    print('WPE per bin: err max %.2e median %.2e; self max %.2e median %.2e; ratio median %.2f'
          % (wpe_err.max(), np.median(wpe_err), wpe_self.max(), np.median(wpe_self), np.median(wpe_err / wpe_self)))
    print('target mask per bin: err max %.2e median %.2e; self max %.2e median %.2e'
          % (m_err.max(), np.median(m_err), m_self.max(), np.median(m_self)))
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
    np.savez(sys.argv[2], err=err, self_f=self_f, cond=cond, nb=nb, wpe_err=wpe_err, wpe_self=wpe_self, m_err=m_err, m_self=m_self)


if __name__ == '__main__':      # the oracle pool spawns workers that re-import this file
    main()
