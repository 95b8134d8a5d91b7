"""The parameter draws of tests/test_gpu_pipeline.py::test_random_shapes_against_oracle, shared
with the replay tools (tools/fuzz_case.py, tools/em_bin_probe.py).  The default stream (seed
2024, not wide) is part of the test suite: do not reorder its draws."""
import os

import numpy as np


def from_environment():
    """(seed, cases, wide) from GSS_FUZZ_SEED / GSS_FUZZ_CASES / GSS_FUZZ_WIDE."""
    return (int(os.environ.get('GSS_FUZZ_SEED', 2024)), int(os.environ.get('GSS_FUZZ_CASES', 40)),
            bool(os.environ.get('GSS_FUZZ_WIDE')))


def fuzz_cases(seed, cases, wide):
    """Yields (case, D, K, N, context_samples, kw) for the runnable draws."""
    rng = np.random.default_rng(seed)
    for case in range(cases):
        D = int(rng.integers(2, 30)); K = int(rng.integers(2, 13) if wide else rng.integers(3, 7))
        N = int(rng.integers(9000, 36000)); ctx_s = int(rng.integers(0, 3000))
        taps = int(rng.integers(1, 4)); delay = int(rng.integers(1, 4)); wit = int(rng.integers(1, 3))
        bss = int(rng.integers(1, 5)); post = int(rng.integers(0, 3))
        bf = ['mvdrSouden_ban', 'ch2', 'sum', 'gev_ban'][int(rng.integers(0, 4 if wide else 3))]
        if bf == 'ch2' and D < 3:
            bf = 'sum'
        pf = [None, 'mask_mul'][int(rng.integers(0, 2))]
        wpe = bool(rng.integers(0, 4) > 0)
        size, shift, fading, psd_context = 1024, 256, True, 0
        if wide:
            # more of the parameter space: other window lengths and overlaps, no fading, a PSD
            # context, longer filters on few channels, now and then up to 19 classes
            size = int([512, 1024, 2048][rng.integers(0, 3)])
            shift = size // int([4, 4, 2][rng.integers(0, 3)])
            fading = bool(rng.integers(0, 5) > 0)
            psd_context = int([0, 0, 1, 4][rng.integers(0, 4)])
            if D <= 8:
                taps = int(rng.integers(1, 7))
            if rng.integers(0, 8) == 0:
                K = int(rng.integers(13, 20))
        pad = size - shift if fading else 0
        T = (N + 2 * pad - size + shift - 1) // shift + 1
        if wpe and T < 3 * taps * D + 10:
            continue                      # too few frames for a well-posed WPE
        kw = dict(wpe=wpe, wpe_taps=taps, wpe_delay=delay, wpe_iterations=wit, bss_iterations=bss,
                  bss_iterations_post=post, bf=bf, postfilter=pf)
        if wide:
            kw.update(stft_size=size, stft_shift=shift, stft_fading=fading,
                      wpe_psd_context=psd_context)
        yield case, D, K, N, ctx_s, kw
