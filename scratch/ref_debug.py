import sys
sys.path.insert(0, 'oracle'); sys.path.insert(0, '.')
import numpy as np
import gss_oracle as oracle
from pb_chime5_amd import ops, synthetic
u = synthetic.config3_item(0)
cs = u.ex['start_orig']['original']; ce = u.ex['end']['original'] - u.ex['end_orig']['original']
x_hat, det = ops.enhance_observation(u.obs, u.activity_array, u.target_index, cs, ce, debug=True)
tm, dm = det['target_mask'], det['distortion_mask']
Yf = det['Obs'].transpose(2, 0, 1)
cov_x = oracle.get_power_spectral_density_matrix(Yf, tm.T)
cov_n = oracle.get_power_spectral_density_matrix(Yf, dm.T)
phi = oracle.stable_solve(cov_n, cov_x)
lam = np.trace(phi, axis1=-1, axis2=-2)[..., None, None]
mat = phi / np.maximum(lam.real, 1e-10)
num = np.einsum('...FdR,...FdD,...FDR->...R', mat.conj(), cov_x, mat).real
den = np.einsum('...FdR,...FdD,...FDR->...R', mat.conj(), cov_n, mat).real
print('finite', np.isfinite(phi).all(), 'nan freq count', int((~np.isfinite(phi).all(axis=(1,2))).sum()))
print('SNR', np.round(num / np.maximum(den, 1e-10) , 4))
print('argmax oracle', int(np.argmax(num / np.maximum(den, 1e-10))), 'gpu', det['ref_channel'])
print('target mask sum', tm.sum(), 'frames with target mass', int((tm.sum(axis=1) > 0).sum()), 'of', tm.shape[0])
mag = np.abs(mat).max(axis=(1, 2))
worst = np.argsort(-mag)[:6]
print('worst bins', worst, mag[worst])
for f in worst[:3]:
    d = dm[:, f]
    print('bin', f, 'dist mask: sum %.3e  nonzero frames %d  max %.3e | target sum %.3e' % (d.sum(), (d > 0).sum(), d.max(), tm[:, f].sum()),
          'cond(cov_n) %.2e' % np.linalg.cond(cov_n[f]))
post = det['posterior']
print('posterior min over classes at worst bin (core frames):', post[:, :, worst[0]].min(axis=1))
