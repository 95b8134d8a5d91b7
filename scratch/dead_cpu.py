import sys
sys.path.insert(0, 'oracle'); sys.path.insert(0, '.')
import numpy as np
import gss_oracle as oracle
from pb_chime5_amd import synthetic
u = synthetic.tiny(seed=3, num_channels=6, num_samples=24000, num_speakers=2)
u.obs[2] = 0.0
Obs = oracle.stft(u.obs)[..., 5:40:7]
X6 = oracle.wpe_block(Obs, 4, 2, 2)
keep = [0, 1, 3, 4, 5]
X5 = oracle.wpe_block(Obs[keep], 4, 2, 2)
rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
print('oracle 6ch(dead) vs 5ch on live channels:', rel(X6[keep], X5), ' dead out max', np.abs(X6[2]).max())
# what happens inside: does solve raise?
Y = Obs[..., 0]
Yt = oracle.build_y_tilde(Y, 4, 2)
ip = oracle.get_power_inverse(Y)
R = (Yt * ip) @ Yt.conj().T
P = (Yt * ip) @ Y.conj().T
try:
    G = np.linalg.solve(R, P); print('solve did not raise; |G| max', np.abs(G).max())
except np.linalg.LinAlgError as e:
    print('raised', e)
G2 = np.linalg.lstsq(R, P, rcond=None)[0]
print('lstsq |G| max', np.abs(G2).max(), 'rank', np.linalg.matrix_rank(R), 'of', R.shape[0])
