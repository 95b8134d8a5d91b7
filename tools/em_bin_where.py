"""Where do the GPU's and the oracle's posteriors of a saved bin (tools/em_bin_probe.py) differ?
    python tools/em_bin_where.py gpurun_out/em_bin_220_43.npz ITER POST"""
import sys
from pathlib import Path

import numpy as np

R = Path(__file__).resolve().parents[1]
for p in (str(R), str(R / 'oracle'), str(R / 'tests')):
    sys.path.insert(0, p)
import gss_oracle as oracle
from pb_chime5_amd import ops

d = np.load(sys.argv[1])
it, post = int(sys.argv[2]), int(sys.argv[3])
Of, act = d['Of'], d['act']
g = ops.cacgmm_posteriors(Of, act, it, post)[..., 0]
o = oracle.gss_block_batched(Of, act, iterations=it, iterations_post=post)[..., 0]
diff = np.abs(g - o)
print('max diff', diff.max(), 'frames with diff > 1e-6:', np.flatnonzero(diff.max(axis=0) > 1e-6))
for t in np.argsort(diff.max(axis=0))[::-1][:8]:
    print(f't={t} act={act[:, t].astype(int)} gpu={np.array2string(g[:, t], precision=6)} '
          f'oracle={np.array2string(o[:, t], precision=6)}')
# scale the observation of single frames: posteriors must not depend on per-frame scale
