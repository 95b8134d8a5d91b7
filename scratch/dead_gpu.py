import sys
sys.path.insert(0, 'oracle'); sys.path.insert(0, '.')
import numpy as np
import gss_oracle as oracle
from pb_chime5_amd import ops, synthetic
u = synthetic.tiny(seed=3, num_channels=6, num_samples=24000, num_speakers=2)
u.obs[2] = 0.0
Obs = oracle.stft(u.obs)
rel = lambda a, b: np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)
for its in (1, 2):
    Xo = oracle.wpe_block(Obs, 4, 2, its)
    Xg = ops.wpe_dtf(Obs, taps=4, delay=2, iterations=its)
    per_f = np.array([rel(Xg[..., f], Xo[..., f]) for f in range(Obs.shape[-1])])
    print('iterations', its, 'overall', rel(Xg, Xo), 'finite', np.isfinite(Xg).all(),
          'bad bins', int((per_f > 1e-6).sum()), 'worst', per_f.max(), 'at', per_f.argmax())
    print('  per channel', [f'{rel(Xg[d], Xo[d]):.1e}' for d in range(6)])
