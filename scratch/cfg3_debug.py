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
cond = np.linalg.cond(cov_n)
w = oracle.get_mvdr_vector_souden(cov_x, cov_n, ref_channel=det['ref_channel'], eps=1e-10)
w = oracle.blind_analytic_normalization(w, cov_n)
X_want = oracle.apply_beamforming_vector(w, Yf).T
err = np.linalg.norm(np.abs(det['X_hat']) - np.abs(X_want), axis=0) / np.maximum(np.linalg.norm(np.abs(X_want), axis=0), 1e-300)
for lo, hi in [(0, 1e4), (1e4, 1e6), (1e6, 1e8), (1e8, 1e10), (1e10, 1e12), (1e12, 1e30)]:
    sel = (cond >= lo) & (cond < hi)
    if sel.any():
        print(f'cond in [{lo:.0e},{hi:.0e}): {sel.sum():3d} bins, max rel err {err[sel].max():.2e}, median {np.median(err[sel]):.2e}')
condx = np.linalg.cond(cov_x)
print('ref', det['ref_channel'])

# --- who is closer to an extended-precision evaluation on the sensitive bins?
L = np.clongdouble
def truth_bin(f):
    Y = Yf[f].astype(L)                                    # (D, T)
    def psd(mask):
        m = mask[:, f].astype(np.longdouble)
        m = m / max(m.sum(), np.longdouble(1e-10))
        return (Y * m) @ Y.conj().T
    cx, cn = psd(tm), psd(dm)
    cn64 = cn.astype(np.complex128)
    phi = np.linalg.solve(cn64, cx.astype(np.complex128)).astype(L)
    for _ in range(8):
        res = cx - cn @ phi
        phi = phi + np.linalg.solve(cn64, res.astype(np.complex128)).astype(L)
    lam = np.trace(phi).real
    w = phi[:, det['ref_channel']] / max(lam, np.longdouble(1e-10))
    nom = np.sqrt(np.abs(w.conj() @ cn @ cn @ w)); den = np.abs(w.conj() @ cn @ w)
    w = w * (nom / den)
    return (w.conj() @ Y).astype(np.complex128)            # (T,)
band = np.flatnonzero((cond >= 1e8) & (cond < 1e10))
pick = band[np.argsort(-err[band])[:6]]
for f in pick:
    xt = truth_bin(f)
    e_gpu = np.linalg.norm(np.abs(det['X_hat'][:, f]) - np.abs(xt)) / np.linalg.norm(np.abs(xt))
    e_orc = np.linalg.norm(np.abs(X_want[:, f]) - np.abs(xt)) / np.linalg.norm(np.abs(xt))
    print(f'bin {f}: cond {cond[f]:.1e}  GPU vs truth {e_gpu:.2e}   oracle(f64) vs truth {e_orc:.2e}')

f = int(pick[0])
Y = Yf[f].astype(L)
def psdL(mask):
    m = mask[:, f].astype(np.longdouble); m = m / max(m.sum(), np.longdouble(1e-10))
    return (Y * m) @ Y.conj().T
cxL, cnL = psdL(tm), psdL(dm)
rel = lambda a, b: float(np.linalg.norm((a - b).astype(np.complex128)) / np.linalg.norm(np.asarray(b).astype(np.complex128)))
print('bin', f, 'cov_x f64 vs long', rel(cov_x[f], cxL), 'cov_n', rel(cov_n[f], cnL))
phi64 = np.linalg.solve(cov_n[f], cov_x[f])
phiL = phi64.astype(L)
for _ in range(8):
    phiL = phiL + np.linalg.solve(cov_n[f], (cxL - cnL @ phiL).astype(np.complex128)).astype(L)
print('phi f64 vs refined', rel(phi64, phiL), ' |phi| max', np.abs(phi64).max(), ' trace', np.trace(phi64), np.trace(phiL))
ev = np.linalg.eigvalsh(cov_n[f]); print('eig cov_n min/max', ev[0], ev[-1], ' eig cov_x max', np.linalg.eigvalsh(cov_x[f])[-1])
print('Hermitian defect cov_n', np.abs(cov_n[f] - cov_n[f].conj().T).max())
