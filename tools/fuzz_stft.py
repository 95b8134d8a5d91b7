"""Random STFT / iSTFT parameter sweep against the oracle (window lengths that are and are not
powers of two, any shift that the biorthogonal synthesis window allows, short and empty signals,
with and without fading):  python tools/fuzz_stft.py [SEED] [CASES]"""
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
    cases = int(sys.argv[2]) if len(sys.argv) > 2 else 150
    rng = np.random.default_rng(seed)
    bad = done = 0
    for case in range(cases):
        if rng.integers(0, 2):
            size = int(2 ** rng.integers(4, 13))
        else:
            size = int(2 * rng.integers(8, 1500))
        divisors = [d for d in (2, 3, 4, 5, 6, 8, 16) if size % d == 0]
        # (the library and nara_wpe's istft want a shift that divides the window)
        shift = size // int(rng.choice(divisors)) if divisors else size // 2
        fading = bool(rng.integers(0, 2))
        N = int(rng.choice([0, 1, size - 1, size, size + 1, int(rng.integers(1, 40000))]))
        D = int(rng.integers(1, 5))
        x = rng.standard_normal((D, N))
        tag = dict(case=case, size=size, shift=shift, fading=fading, N=N, D=D)
        try:
            want = oracle.stft(x, size, shift, fading=fading)
        except Exception as e:                       # the reference's own limits
            try:
                ops.stft(x, size, shift, fading=fading)
                print('oracle raises', type(e).__name__, 'GPU does not:', tag)
                bad += 1
            except Exception:
                pass
            continue
        try:
            got = ops.stft(x, size, shift, fading=fading)
        except Exception as e:
            print('GPU raises', type(e).__name__, str(e)[:100], tag)
            bad += 1
            continue
        done += 1
        if got.shape != want.shape:
            print('shape', got.shape, want.shape, tag)
            bad += 1
            continue
        scale = max(np.abs(want).max(), 1e-300) if want.size else 1.0
        if want.size and np.abs(got - want).max() > 1e-10 * scale:
            print('stft', np.abs(got - want).max() / scale, tag)
            bad += 1
        if want.shape[1] == 0:
            continue
        X = want
        try:
            xo = oracle.istft(X, size, shift, fading=fading)
        except Exception:
            continue
        try:
            xg = ops.istft(X, size, shift, fading=fading)
        except Exception as e:
            print('GPU istft raises', type(e).__name__, str(e)[:100], tag)
            bad += 1
            continue
        # (a one-sample signal without fading meets the zero at the start of the window: the
        # output is 1e-35 and only absolute agreement means anything)
        scale = max(np.abs(xo).max(), 1e-12 * (np.abs(x).max() if x.size else 0.0), 1e-300) if xo.size else 1.0
        if xg.shape != xo.shape or (xo.size and not np.allclose(xg, xo, rtol=0, atol=1e-9 * scale)):
            err = np.abs(xg - xo).max() / scale if xg.shape == xo.shape and xo.size else None
            print('istft', xg.shape, xo.shape, err, tag)
            bad += 1
    print('stft fuzz: seed', seed, 'compared', done, 'of', cases, 'failures', bad)


if __name__ == '__main__':
    main()
