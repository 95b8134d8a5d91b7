"""How much does the ORACLE's own WPE output move when its input is perturbed in the
last bit?  (conditioning floor of any GPU-vs-CPU comparison on this data)"""
import sys
import numpy as np
sys.path.insert(0, 'oracle'); sys.path.insert(0, '.')
import gss_oracle as oracle
from pb_chime5_amd import synthetic
utt = synthetic.make_utterance(3, 8, 128000, [(8000, 120000), (0, 70000), (50000, 128000)], target=0,
                               start_context=8000, end_context=8000, rir_taps=1024)
Obs = oracle.stft(utt.obs, 1024, 256, fading=True)
sel = slice(0, 513, 16)
Y = Obs[..., sel]
rng = np.random.default_rng(0)
X0 = oracle.wpe_block(Y, 10, 2, 3)
for eps in (1.1e-16, 1e-14, 1e-12):
    Yp = Y * (1.0 + eps * rng.standard_normal(Y.shape))
    X1 = oracle.wpe_block(Yp, 10, 2, 3)
    print(f'input perturbation {eps:.1e} -> output change {np.linalg.norm(X1 - X0) / np.linalg.norm(X0):.2e}')
