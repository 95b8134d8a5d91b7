"""Random WPE shapes against the oracle: channel counts, filter lengths and delays whose
taps * D lands on and around the block boundaries of the Cholesky solve (48, 96, ... and the 16-row
panels inside), frame counts around the 64-frame chunks of the correlation kernel, one to three
iterations, with and without a PSD context:  python tools/fuzz_wpe.py [SEED] [CASES]"""
import os
import sys
from pathlib import Path

import numpy as np

R = Path(__file__).resolve().parents[1]
for p in (str(R), str(R / 'oracle'), str(R / 'tests')):
    sys.path.insert(0, p)


def main():
    import gss_oracle as oracle
    from pb_chime5_amd import ops
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    cases = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    rng = np.random.default_rng(seed)
    bad = 0
    worst = 0.0
    ctx = ops.default_context()
    for case in range(cases):
        D = int(rng.integers(1, 31))
        if os.environ.get('GSS_FUZZ_D'):      # aim at one correlation kernel family
            D = int(os.environ['GSS_FUZZ_D'])
        taps = int(rng.integers(1, 11))
        if rng.integers(0, 3) == 0:       # aim at a block boundary
            target = int(rng.choice([16, 32, 48, 64, 96, 144, 192, 240, 288])) + int(rng.integers(-1, 2))
            taps = max(1, min(10, round(target / D)))
        n = taps * D
        if n > 300:
            continue
        delay = int(rng.integers(1, 4))      # (delay 0 predicts every frame from itself: X = 0)
        iterations = int(rng.integers(1, 4))
        psd = int([0, 0, 0, 2][rng.integers(0, 4)])
        T = int(max(2 * n + 20, rng.integers(40, 400)) + rng.integers(0, 70))
        F = int(rng.integers(1, 6))
        # a reverberant toy signal per frequency, so that the filter has something to remove
        S = rng.standard_normal((F, D, T + 12)) + 1j * rng.standard_normal((F, D, T + 12))
        h = 0.6 ** np.arange(12)
        Y = sum(h[k] * S[..., 12 - k:12 - k + T] for k in range(12))
        Y = np.ascontiguousarray(Y.transpose(1, 2, 0))                 # (D, T, F)
        tag = dict(case=case, D=D, taps=taps, n=n, delay=delay, iterations=iterations, psd=psd, T=T, F=F)
        want = oracle.wpe_v8(Y.transpose(2, 0, 1), taps, delay, iterations, psd).transpose(1, 2, 0)
        try:
            got = ops.wpe_dtf(Y, taps, delay, iterations, psd_context=psd, ctx=ctx)
        except Exception as e:
            print('GPU raises', type(e).__name__, str(e)[:120], tag)
            bad += 1
            continue
        err = np.linalg.norm(got - want) / np.linalg.norm(want)
        worst = max(worst, err)
        if not err < 1e-8 or ctx.last_wpe_zero_pivots() != 0:
            print('wpe', err, 'zero pivots', ctx.last_wpe_zero_pivots(), tag)
            bad += 1
    print('wpe fuzz: seed', seed, 'cases', cases, 'failures', bad, 'worst', worst)


if __name__ == '__main__':
    main()
