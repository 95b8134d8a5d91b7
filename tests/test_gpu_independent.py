"""The HIP stages against the INDEPENDENT implementations of tests/test_oracle_independent.py
(scipy's STFT / iSTFT, WPE as a direct least-squares problem, the guided EM and the
beamformers by brute-force per-frequency loops) -- without the oracle in between.  Through
the C ABI like every other GPU test."""
import numpy as np
import pytest
import scipy.linalg
import scipy.signal

from conftest import rel_err
from test_oracle_independent import (_periodic_blackman, _scene, brute_force_guided_em, crandn,
                                     guided_scene)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('n', [5000, 16001, 80000])
def test_stft_matches_scipy_short_time_fft(gpu_ctx, n):
    from pb_chime5_amd import ops
    rng = np.random.default_rng(n)
    x = rng.standard_normal((2, n))
    size, shift = 1024, 256
    got = ops.stft(x, size, shift, ctx=gpu_ctx)                         # (2, T, F)
    T = got.shape[1]
    pad = size - shift
    sft = scipy.signal.ShortTimeFFT(_periodic_blackman(size), hop=shift, fs=1.0,
                                    fft_mode='onesided', scale_to=None, phase_shift=None)
    for d in range(2):
        xp = np.zeros((T - 1) * shift + size)
        xp[pad:pad + n] = x[d]
        off = size // (2 * shift)
        want = sft.stft(xp, p0=off, p1=off + T).T
        assert rel_err(got[d], want) < 1e-12


@pytest.mark.parametrize('T', [20, 316])
def test_istft_matches_scipy_least_squares_synthesis(gpu_ctx, T):
    from pb_chime5_amd import ops
    rng = np.random.default_rng(T)
    size, shift = 1024, 256
    X = crandn(rng, T, size // 2 + 1)
    X[:, 0] = X[:, 0].real
    X[:, -1] = X[:, -1].real
    got = ops.istft(X, size, shift, ctx=gpu_ctx)
    w = _periodic_blackman(size)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        _, x = scipy.signal.istft(X.T / w.sum(), window=w, nperseg=size, noverlap=size - shift,
                                  input_onesided=True, boundary=False)
    pad = size - shift
    want = x[pad:pad + got.shape[0]]
    inner = slice(shift, got.shape[0] - shift)
    assert rel_err(got[inner], want[inner]) < 1e-10      # nara_wpe's window quirk: 4e-12


@pytest.mark.parametrize('D,T,taps,delay', [(2, 120, 3, 2), (4, 300, 5, 3), (6, 400, 4, 1)])
def test_wpe_is_the_weighted_least_squares_solution(gpu_ctx, D, T, taps, delay):
    """Two WPE iterations on the GPU against np.linalg.lstsq on the sqrt(w)-scaled regressor
    matrix (no normal equations), weights recomputed from the lstsq result in between."""
    from pb_chime5_amd import ops
    rng = np.random.default_rng(D * T)
    F = 3
    Ys = []
    wants = []
    for f in range(F):
        S = crandn(rng, 1, T + 8)
        h = crandn(rng, D, 8) * np.exp(-np.arange(8))
        Y = sum(h[:, k:k + 1] * S[:, 8 - k:8 - k + T] for k in range(8)) + 0.05 * crandn(rng, D, T)
        A = np.zeros((T, taps * D), complex)
        for t in range(T):
            for j in range(taps):
                src = t - (delay + taps - 1 - j)
                if src >= 0:
                    A[t, j * D:(j + 1) * D] = Y[:, src].conj()
        X = Y
        for _ in range(2):
            power = np.mean(np.abs(X) ** 2, axis=0)
            sw = np.sqrt(1 / np.maximum(power, 1e-10 * power.max()))[:, None]
            G = np.linalg.lstsq(sw * A, sw * Y.conj().T, rcond=None)[0]
            X = Y - (A @ G).conj().T
        Ys.append(Y)
        wants.append(X)
    got = ops.wpe_dtf(np.stack(Ys, axis=-1), taps, delay, 2, ctx=gpu_ctx)
    assert rel_err(got, np.stack(wants, axis=-1)) < 1e-8


@pytest.mark.parametrize('D,K,iterations,post', [(4, 3, 5, 1), (6, 4, 4, 0), (24, 5, 3, 1), (5, 3, 3, 3)])
def test_guided_em_by_brute_force(gpu_ctx, D, K, iterations, post):
    from pb_chime5_amd import ops
    rng = np.random.default_rng(100 * D + 10 * iterations + post)
    F = 3
    T = 110 if D < 20 else 160
    obs, acts = [], None
    for f in range(F):
        o, a = guided_scene(np.random.default_rng(7), K=K, D=D, T=T)      # same activity ...
        acts = a
        obs.append(o + 0.1 * crandn(rng, T, D))                            # ... different signals
    Obs = np.stack(obs, axis=-1).transpose(1, 0, 2)                        # (D, T, F)
    got = ops.cacgmm_posteriors(Obs, acts, iterations, post, ctx=gpu_ctx)  # (K, T, F)
    for f in range(F):
        want = brute_force_guided_em(obs[f], acts, iterations, post)
        # (the brute-force side inverts 24 x 24 covariance matrices of 160 frames: 2e-8)
        assert np.max(np.abs(got[..., f] - want)) < 1e-6, f


def test_mvdr_souden_ban_and_gev_by_per_frequency_loops(gpu_ctx):
    from pb_chime5_amd import ops
    rng = np.random.default_rng(5)
    Y, mx, mn = _scene(rng, F=7, D=6, T=120)
    F, D, T = Y.shape
    cov_x, cov_n, mats = [], [], []
    for f in range(F):
        cx = sum(mx[f, t] * np.outer(Y[f, :, t], Y[f, :, t].conj()) for t in range(T)) / mx[f].sum()
        cn = sum(mn[f, t] * np.outer(Y[f, :, t], Y[f, :, t].conj()) for t in range(T)) / mn[f].sum()
        phi = scipy.linalg.solve(cn, cx)
        cov_x.append(cx)
        cov_n.append(cn)
        mats.append(phi / max(np.trace(phi).real, 1e-10))
    snr = [sum(np.real(mats[f][:, d].conj() @ cov_x[f] @ mats[f][:, d]) for f in range(F)) /
           sum(np.real(mats[f][:, d].conj() @ cov_n[f] @ mats[f][:, d]) for f in range(F))
           for d in range(D)]
    ref = int(np.argmax(snr))
    Ydtf = Y.transpose(1, 2, 0)
    X, got_ref = ops.mvdr_souden_from_masks(Ydtf, mx.T, mn.T, ban=True, return_ref_channel=True,
                                            ctx=gpu_ctx)
    assert got_ref == ref
    Xg = ops.gev_from_masks(Ydtf, mx.T, mn.T, ban=True, ctx=gpu_ctx)
    for f in range(F):
        wf = mats[f][:, ref]
        pw = cov_n[f] @ wf
        w_ban = wf * np.linalg.norm(pw) / abs(np.vdot(wf, pw))
        assert rel_err(X[:, f], w_ban.conj() @ Y[f]) < 1e-10
        v = scipy.linalg.eigh(cov_x[f], cov_n[f])[1][:, -1]
        pv = cov_n[f] @ v
        v_ban = v * np.linalg.norm(pv) / abs(np.vdot(v, pv))
        assert rel_err(np.abs(Xg[:, f]), np.abs(v_ban.conj() @ Y[f])) < 1e-9
