import sys
sys.path.insert(0, 'oracle'); sys.path.insert(0, '.')
import numpy as np
import gss_oracle as oracle
from pb_chime5_amd import ops, synthetic
u = synthetic.tiny(seed=3, num_channels=6, num_samples=24000, num_speakers=2)
u.obs[2] = 0.0     # a dead microphone
cs = u.ex['start_orig']['original']
kw = dict(wpe=True, wpe_taps=4, wpe_delay=2, wpe_iterations=2, bss_iterations=5)
got, det = ops.enhance_observation(u.obs, u.activity_array, u.target_index, cs, cs, debug=True, **kw)
want, wdet = oracle.enhance_observation(u.obs, u.activity_array, u.target_index, u.ex, return_details=True, **kw)
rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
print('finite', np.isfinite(got).all(), np.isfinite(want).all())
print('Obs', rel(det['Obs'], wdet['Obs']), 'dead ch max', np.abs(det['Obs'][2]).max(), np.abs(wdet['Obs'][2]).max())
print('x_hat', rel(got, want))
