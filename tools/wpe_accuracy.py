"""WPE accuracy on the bench workload: GPU and oracle against the extended-precision solution
of tests/golden/wpe_truth_config2.npz (3 bins of config 2; tests/golden/make_wpe_truth.py).

    python tools/wpe_accuracy.py

Prints, per bin and iteration count, ||X - X_ext|| / ||X_ext|| for the oracle (float64, LU)
and for the HIP path, their ratio, and GPU vs oracle directly."""
import sys
from pathlib import Path

import numpy as np

R = Path(__file__).resolve().parents[1]
for p in (str(R), str(R / 'oracle')):
    if p not in sys.path:
        sys.path.insert(0, p)

import gss_oracle as oracle  # noqa: E402
from pb_chime5_amd import ops  # noqa: E402

g = np.load(R / 'tests' / 'golden' / 'wpe_truth_config2.npz')
Y, taps, delay = g['Y'], int(g['taps']), int(g['delay'])
n = np.linalg.norm
for iters, key in ((1, 'X1'), (3, 'X3')):
    got = ops.wpe_dtf(Y, taps, delay, iters)
    want = oracle.wpe_block(Y, taps, delay, iters)
    for i, f in enumerate(g['bins']):
        t = g[key][..., i]
        e_or = n(want[..., i] - t) / n(t)
        e_gpu = n(got[..., i] - t) / n(t)
        print(f'iterations {iters} bin {int(f):3d}: oracle {e_or:.2e}  gpu {e_gpu:.2e}  '
              f'ratio {e_gpu / e_or:5.2f}   gpu vs oracle {n(got[..., i] - want[..., i]) / n(t):.2e}')
from pb_chime5_amd._capi import default_context  # noqa: E402
print('zeroed pivots:', default_context().last_wpe_zero_pivots())
