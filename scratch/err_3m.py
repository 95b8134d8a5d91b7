"""Relative error of the enhanced signal against the oracle, 4M vs 3M correlation.
Run with GSS_CORR_4M unset (3-product form, default) / set (4-product form)."""
import os, sys, time
import numpy as np
sys.path.insert(0, 'oracle'); sys.path.insert(0, '.')
import gss_oracle as oracle
from pb_chime5_amd import synthetic, ops
from pb_chime5_amd._capi import default_context

utt = synthetic.make_utterance(3, 8, 128000, [(8000, 120000), (0, 70000), (50000, 128000)], target=0,
                               start_context=8000, end_context=8000, rir_taps=1024)
ctx = default_context(0)
params = ops.make_params(wpe=True, wpe_taps=10, wpe_delay=2, wpe_iterations=3, bss_iterations=20,
                         bss_iterations_post=1)
x_gpu = ops.enhance_observation(utt.obs, utt.activity_array, utt.target_index, 8000, 8000, params=params, ctx=ctx)
t = time.time()
x_ref = oracle.enhance_observation(utt.obs, utt.activity_array, utt.target_index, utt.ex, wpe=True, wpe_taps=10,
                                   wpe_delay=2, wpe_iterations=3, bss_iterations=20, bss_iterations_post=1)
print('oracle s', round(time.time() - t, 1))
e = np.linalg.norm(x_gpu - x_ref) / np.linalg.norm(x_ref)
print('4M' if os.environ.get('GSS_CORR_4M') else '3M', 'rel err', e)
