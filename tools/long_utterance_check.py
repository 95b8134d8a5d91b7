"""A 300 s, 24-channel utterance (T = 18750 frames, 2.3e8 STFT bins: 3.7 GB per copy) through the
fused pipeline, and three of its frequencies through the stage operators on their own: the same
kernels at F = 513 and F = 3 must give the same bits unless an index overflows somewhere.
    python tools/long_utterance_check.py [SECONDS]"""
import sys
import time
from pathlib import Path

import numpy as np

R = Path(__file__).resolve().parents[1]
for p in (str(R), str(R / 'oracle')):
    sys.path.insert(0, p)


def main():
    from pb_chime5_amd import ops, synthetic
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
    N = int(seconds * 16000)
    u = synthetic.tiny(seed=3, num_channels=24, num_samples=N, num_speakers=4, context=16000, noise=3e-2)
    kw = dict(wpe=True, wpe_taps=10, wpe_delay=2, wpe_iterations=2, bss_iterations=3,
              bss_iterations_post=1, bf='mvdrSouden_ban')
    cs = u.ex['start_orig']['original']
    t0 = time.perf_counter()
    x, det = ops.enhance_observation(u.obs, u.activity_array, u.target_index, cs, cs, debug=True, **kw)
    print('frames', det['Obs'].shape[1], 'fused pipeline %.2f s' % (time.perf_counter() - t0),
          'finite', bool(np.all(np.isfinite(x))), 'workspace GB',
          ops.default_context().workspace_bytes() / 1e9 if hasattr(ops.default_context(), 'workspace_bytes') else '?')
    bins = [0, 257, 512]
    Y = ops.stft(u.obs)[..., bins]                                     # (D, T, 3)
    X = ops.wpe_dtf(Y, 10, 2, 2)
    print('WPE of bins', bins, 'alone vs inside the full run: max abs diff',
          np.max(np.abs(X - det['Obs'][..., bins])))
    act = det['acitivity_freq']
    post = ops.cacgmm_posteriors(det['Obs'][..., bins], act, 3, 1)
    print('EM alone vs inside: max abs diff', np.max(np.abs(post - det['posterior'][..., bins])))


if __name__ == '__main__':
    main()
