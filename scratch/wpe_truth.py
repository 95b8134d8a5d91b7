"""One WPE iteration on a few frequencies: GPU and f64 oracle against an extended-
precision solution (R, P formed and the solve refined in 80-bit long double)."""
import sys
import numpy as np
sys.path.insert(0, 'oracle'); sys.path.insert(0, '.')
import gss_oracle as oracle
from pb_chime5_amd import synthetic

D = int(sys.argv[1]) if len(sys.argv) > 1 else 8
utt = synthetic.make_utterance(3, D, 128000, [(8000, 120000), (0, 70000), (50000, 128000)], target=0,
                               start_context=8000, end_context=8000, rir_taps=1024)
Obs = oracle.stft(utt.obs, 1024, 256, fading=True)          # (D, T, F)
bins = list(range(5, 513, 64))
Y = np.ascontiguousarray(Obs[..., bins])                      # (D, T, Fs)
taps, delay = 10, 2

def truth(Yf):
    Yt = oracle.build_y_tilde(Yf, taps, delay)
    ip = oracle.get_power_inverse(Yf, psd_context=0)
    L = np.clongdouble
    Ytl, Yl = Yt.astype(L), Yf.astype(L)
    A = Ytl * ip.astype(np.longdouble)[None, :]
    R = A @ Ytl.conj().T
    P = A @ Yl.conj().T
    R64 = R.astype(np.complex128)
    G = np.linalg.solve(R64, P.astype(np.complex128)).astype(L)
    for _ in range(6):
        res = P - R @ G
        G = G + np.linalg.solve(R64, res.astype(np.complex128)).astype(L)
    X = Yl - G.conj().T @ Ytl
    cond = np.linalg.cond(R64)
    return X.astype(np.complex128), cond

Xt = np.empty_like(Y)
conds = []
for i in range(len(bins)):
    Xt[..., i], c = truth(Y[..., i])
    conds.append(c)
Xo = oracle.wpe_block(Y, taps, delay, 1)
rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
print('cond(R) per bin', ' '.join(f'{c:.1e}' for c in conds))
print('oracle f64 vs truth', rel(Xo, Xt))
try:
    from pb_chime5_amd import ops
    from pb_chime5_amd._capi import default_context
    Xg = ops.wpe_dtf(Y, taps=taps, delay=delay, iterations=1, ctx=default_context(0))
    print('GPU        vs truth', rel(Xg, Xt))
    print('GPU        vs oracle', rel(Xg, Xo))
except Exception as e:
    print('no GPU:', e)
for its in (2, 3):
    Xo = oracle.wpe_block(Y, taps, delay, its)
    try:
        Xg = ops.wpe_dtf(Y, taps=taps, delay=delay, iterations=its, ctx=default_context(0))
        print(f'iterations={its}: GPU vs oracle', rel(Xg, Xo), ' per bin',
              ' '.join(f'{rel(Xg[..., i], Xo[..., i]):.1e}' for i in range(len(bins))))
    except Exception as e:
        print(e)
