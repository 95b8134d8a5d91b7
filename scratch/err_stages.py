"""Stage-wise GPU-vs-oracle differences on a mid-size utterance (8 ch, 8 s)."""
import os, sys
import numpy as np
sys.path.insert(0, 'oracle'); sys.path.insert(0, '.')
import gss_oracle as oracle
from pb_chime5_amd import synthetic, ops
from pb_chime5_amd._capi import default_context

utt = synthetic.make_utterance(3, 8, 128000, [(8000, 120000), (0, 70000), (50000, 128000)], target=0,
                               start_context=8000, end_context=8000, rir_taps=1024)
ctx = default_context(0)
kw = dict(wpe=True, wpe_taps=10, wpe_delay=2, wpe_iterations=3, bss_iterations=20, bss_iterations_post=1)
params = ops.make_params(**kw)
x_gpu, d = ops.enhance_observation(utt.obs, utt.activity_array, utt.target_index, 8000, 8000, params=params,
                                   ctx=ctx, debug=True)
x_ref, r = oracle.enhance_observation(utt.obs, utt.activity_array, utt.target_index, utt.ex,
                                      return_details=True, **kw)
rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
print('Obs (after WPE)   ', rel(d['Obs'], r['Obs']))
print('posterior         ', rel(d['posterior'], r['masks'] if r['masks'].shape == d['posterior'].shape else r['masks']), np.abs(d['posterior'] - r['masks']).max())
print('X_hat             ', rel(d['X_hat'], r['X_hat']), ' |X_hat| ', rel(np.abs(d['X_hat']), np.abs(r['X_hat'])))
print('ref channel       ', d['ref_channel'], r.get('ref_channel'))
print('x_hat             ', rel(x_gpu, x_ref))
# GPU back half on the oracle's dereverberated STFT: isolates the EM + MVDR sensitivity
post = ops.cacgmm_posteriors(r['Obs'], r['activity_freq'], iterations=20, iterations_post=1, ctx=ctx)
print('posterior | oracle Obs', rel(post, oracle.gss_block(r['Obs'], r['activity_freq'], iterations=20, iterations_post=1)))
