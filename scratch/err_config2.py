"""GPU vs oracle on the full bench workload (config 2) -- stage-wise."""
import sys, time
import numpy as np
sys.path.insert(0, 'oracle'); sys.path.insert(0, '.')
import gss_oracle as oracle
from pb_chime5_amd import synthetic, ops
from pb_chime5_amd._capi import default_context
utt = synthetic.config2(seed=2)
ctx_s = utt.ex['start_orig']['original']
kw = dict(wpe=True, wpe_taps=10, wpe_delay=2, wpe_iterations=3, bss_iterations=20, bss_iterations_post=1)
x_gpu, d = ops.enhance_observation(utt.obs, utt.activity_array, utt.target_index, ctx_s, ctx_s,
                                   params=ops.make_params(**kw), ctx=default_context(0), debug=True)
t = time.time()
x_ref, r = oracle.enhance_observation(utt.obs, utt.activity_array, utt.target_index, utt.ex,
                                      return_details=True, **kw)
print('oracle seconds', round(time.time() - t, 1))
rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
print('Obs after WPE', rel(d['Obs'], r['Obs']))
per_bin = np.linalg.norm(d['Obs'] - r['Obs'], axis=(0, 1)) / np.linalg.norm(r['Obs'], axis=(0, 1))
print('  per-bin: median %.1e  p90 %.1e  max %.1e (bin %d)' % (np.median(per_bin), np.quantile(per_bin, .9), per_bin.max(), per_bin.argmax()))
print('|X_hat|', rel(np.abs(d['X_hat']), np.abs(r['X_hat'])), ' X_hat', rel(d['X_hat'], r['X_hat']))
print('x_hat', rel(x_gpu, x_ref))
np.savez_compressed('gpurun_out/config2_per_bin.npz', per_bin=per_bin)
