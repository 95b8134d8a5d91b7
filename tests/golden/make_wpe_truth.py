"""Extended-precision WPE on a few frequency bins of the bench workload (BASELINE.json
configs[1]: 24 channels, T = 941, 10 taps, delay 2, 3 iterations) ->
tests/golden/wpe_truth_config2.npz.

    python tests/golden/make_wpe_truth.py

What it pins.  nara_wpe.wpe.wpe_v6 (call site /root/reference/pb_chime5/core.py:48-58) is a
weighted least-squares problem per frequency and iteration.  Its exact solution does not
depend on how the normal equations are formed or solved; float64 implementations (the
reference's einsum + np.linalg.solve, the oracle's restatement, the HIP kernels) scatter
around it by cond(R) * eps.  This script evaluates the same iteration in 80-bit extended
precision (numpy longdouble: powers, weights, correlation matrices, a Cholesky solve with
one refinement step, the filter application) from the float64 STFT of the seeded
utterance.  The tests then hold the GPU to the ORACLE'S OWN distance from that solution
(tests/test_gpu_stages.py::test_wpe_config2_bins_within_oracle_noise_of_extended_precision),
which is the statement "inside the reference's rounding noise" without a tolerance that
has to be guessed per scene.

The fixture is data: the float64 input bins Y and the extended-precision outputs after 1
and 3 iterations, rounded to float64.
"""
import sys
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parents[2]
for p in (str(REPO), str(REPO / 'oracle'), str(REPO / 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)

BINS = (169, 40, 300)        # 169: the worst bin of the end-to-end test in rounds 1-2
TAPS, DELAY = 10, 2


def wpe_ld(Y, iterations):
    """Extended-precision WPE of one frequency (tests/ext_precision.py): Y (D, T) complex128 ->
    list of X after every iteration, complex128."""
    import ext_precision
    import gss_oracle as oracle
    return ext_precision.wpe(Y, oracle.build_y_tilde(Y, TAPS, DELAY), iterations)


def _one(args):
    Yf, iterations = args
    return wpe_ld(Yf, iterations)


def main():
    import multiprocessing as mp
    import gss_oracle as oracle
    from pb_chime5_amd import synthetic
    u = synthetic.config2()
    Y = np.ascontiguousarray(oracle.stft(u.obs)[..., list(BINS)])      # (D, T, nb)
    with mp.Pool(len(BINS)) as pool:
        res = pool.map(_one, [(Y[..., i], 3) for i in range(len(BINS))])
    X1 = np.stack([r[0] for r in res], axis=-1)
    X3 = np.stack([r[2] for r in res], axis=-1)
    out = REPO / 'tests' / 'golden' / 'wpe_truth_config2.npz'
    np.savez_compressed(out, bins=np.array(BINS), taps=TAPS, delay=DELAY, Y=Y, X1=X1, X3=X3)
    for i, f in enumerate(BINS):
        lu1 = oracle.wpe_v6(Y[..., i], TAPS, DELAY, 1)
        lu3 = oracle.wpe_v6(Y[..., i], TAPS, DELAY, 3)
        n = np.linalg.norm
        print(f'bin {f}: oracle (float64, LU) vs extended precision: '
              f'{n(lu1 - X1[..., i]) / n(X1[..., i]):.2e} after 1 iteration, '
              f'{n(lu3 - X3[..., i]) / n(X3[..., i]):.2e} after 3')
    print('wrote', out, out.stat().st_size, 'bytes')


if __name__ == '__main__':
    main()
