import sys
sys.path.insert(0, 'oracle'); sys.path.insert(0, '.')
import numpy as np
import gss_oracle as oracle
from pb_chime5_amd import ops
rng = np.random.default_rng(0)
F, T = 6, 260
bad = []
for D in (2, 3, 4, 5, 7, 8, 10, 12, 13, 16, 20, 24, 28):
    for taps in (1, 2, 3, 4, 5, 10):
        n = D * taps
        if n > 300 or T < 3 * n: continue
        Y = rng.standard_normal((D, T, F)) + 1j * rng.standard_normal((D, T, F))
        Y[:, 1:] += 0.7 * Y[:, :-1]; Y[:, 3:] += 0.4 * Y[:, :-3]
        Xo = oracle.wpe_block(Y, taps, 2, 2)
        Xg = ops.wpe_dtf(Y, taps=taps, delay=2, iterations=2)
        e = np.linalg.norm(Xg - Xo) / np.linalg.norm(Xo)
        flag = '  <-- BAD' if e > 1e-9 else ''
        if flag: bad.append((D, taps, n, e))
        print(f'D={D:2d} taps={taps:2d} n={n:3d}  rel err {e:.2e}{flag}')
print('bad:', bad)
