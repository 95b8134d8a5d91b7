"""One case of tests/test_gpu_pipeline.py::test_other_channel_and_class_counts, stage errors."""
import sys
sys.path.insert(0, 'oracle'); sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import gss_oracle as oracle
from pb_chime5_amd import synthetic
import test_gpu_pipeline as tp
D, K = int(sys.argv[1]), int(sys.argv[2])
u = synthetic.tiny(seed=D + K, num_channels=D, num_samples=64000, num_speakers=K - 1, context=4096, noise=3e-2)
got, det, want, wdet = tp._run_both(u, wpe=True, wpe_taps=2, wpe_delay=2, wpe_iterations=2, bss_iterations=6)
rel = tp.rel_err
print('Obs', rel(det['Obs'], wdet['Obs']), 'posterior', np.abs(det['posterior'] - wdet['masks']).max(),
      'ref', det['ref_channel'], wdet['ref_channel'])
e = np.abs(np.abs(det['X_hat']) - np.abs(wdet['X_hat']))
print('|X_hat|', rel(np.abs(det['X_hat']), np.abs(wdet['X_hat'])), 'x_hat', rel(got, want))
cond = np.linalg.cond(wdet['cov_n'])
print('cond(cov_n) median / max', np.median(cond), cond.max())
per_f = np.linalg.norm(e, axis=0) / np.linalg.norm(np.abs(wdet['X_hat']), axis=0)
worst = np.argsort(per_f)[-5:]
print('worst bins', worst, per_f[worst], cond[worst])
dp = np.abs(det['posterior'] - wdet['masks'])
print('posterior shape', dp.shape, 'frac > 1e-3', (dp > 1e-3).mean())
bad_f = np.where((dp > 1e-3).any(axis=(0, 1)))[0] if dp.shape[-1] == cond.shape[0] else None
print('bad bins', None if bad_f is None else (len(bad_f), bad_f[:20]))
print('activity rows sum', u.activity_array.sum(axis=1), u.activity_array.shape)
eo = np.abs(det['Obs'] - wdet['Obs'])
print('Obs abs max', np.abs(wdet['Obs']).max(), 'median', np.median(np.abs(wdet['Obs'])), 'err max', eo.max())
pf = np.linalg.norm(eo, axis=(0, 1)) / np.linalg.norm(wdet['Obs'], axis=(0, 1))
print('Obs err per bin: worst', np.sort(pf)[-5:], 'median', np.median(pf))
print('shapes', det['posterior'].shape, wdet['masks'].shape, 'frac>1e-3 %.4f  frac>1e-6 %.4f' % ((dp > 1e-3).mean(), (dp > 1e-6).mean()))
strict = cond < 1e8
print('bins cond<1e8:', strict.sum(), ' |X_hat| on them:', rel(np.abs(det['X_hat'][:, strict]), np.abs(wdet['X_hat'][:, strict])) if strict.any() else None)
for thr in (1e10, 1e12, 1e14):
    m = cond < thr
    print('cond <', thr, m.sum(), rel(np.abs(det['X_hat'][:, m]), np.abs(wdet['X_hat'][:, m])))
g = ops_post = None
