"""Invariant / closed-form checks of the parts of the oracle that no reference
vector pins: WPE, CACGMM, PSD / MVDR / BAN, iSTFT."""
import numpy as np

import gss_oracle as oracle
from conftest import rel_err


def crandn(rng, *shape):
    return rng.standard_normal(shape) + 1j * rng.standard_normal(shape)


def test_istft_reconstructs_for_any_length():
    rng = np.random.default_rng(0)
    for n in (1000, 4096, 5555):
        x = rng.standard_normal(n)
        y = oracle.istft(oracle.stft(x))
        assert y.shape[0] == oracle.stft_frames(n) * 256 + 768 - 1536
        assert np.max(np.abs(y[:n] - x)) < 1e-9      # 1e-11-level upstream quirk, see oracle
    assert oracle.stft_frames(80000) == 316 and oracle.stft_frames(240000) == 941
    assert oracle.stft_frames(1920000) == 7503


def test_wpe_normal_equations_delay_and_permutation():
    rng = np.random.default_rng(1)
    D, T, taps, delay = 3, 200, 4, 2
    Y = crandn(rng, D, T)
    X = oracle.wpe_v6(Y, taps, delay, iterations=1)
    Yt = oracle.build_y_tilde(Y, taps, delay)
    w = oracle.get_power_inverse(Y)
    R = (Yt * w) @ Yt.conj().T
    P = (Yt * w) @ Y.conj().T
    G = np.linalg.solve(R, P)
    assert np.linalg.norm(R @ G - P) < 1e-9 * np.linalg.norm(P)
    assert rel_err(X, Y - G.conj().T @ Yt) < 1e-12
    # no past available in the first `delay` frames
    assert np.array_equal(X[:, :delay], Y[:, :delay])
    # channel permutation equivariance
    perm = [2, 0, 1]
    assert rel_err(oracle.wpe_v6(Y[perm], taps, delay, 2), oracle.wpe_v6(Y, taps, delay, 2)[perm]) < 1e-9
    # build_y_tilde layout (nara_wpe doctest: T=20, D=2, taps=4, delay=2)
    Yi = np.arange(1, 41).reshape(20, 2).T
    Yti = oracle.build_y_tilde(Yi, 4, 2)
    assert Yti.shape == (8, 20)
    assert list(Yti[0, :8]) == [0, 0, 0, 0, 0, 1, 3, 5]
    assert list(Yti[-2, :5]) == [0, 0, 1, 3, 5]


def test_cacgmm_posteriors_sum_mask_and_monotone_likelihood():
    rng = np.random.default_rng(2)
    D, T, K = 4, 150, 3
    act = np.zeros((K, T), bool)
    act[0, 10:90] = True
    act[1, 60:140] = True
    act[2] = True
    y = 0.2 * crandn(rng, T, D)
    for k in range(2):
        y += (crandn(rng, 1, D) * crandn(rng, T, 1)) * act[k][:, None]
    init, mask = oracle.gss_initialization(act)
    trainer = oracle.CACGMMTrainer()
    yn = oracle.normalize_observation(y)
    prev = -np.inf
    for it in range(1, 8):
        model = trainer.fit(y, init, iterations=it, source_activity_mask=mask)
        aff, _ = model._predict(yn, source_activity_mask=mask, affiliation_eps=1e-10)
        assert np.all(aff[~mask] == 1e-10)             # masked-out classes sit on the clip floor
        log_pdf, _ = model._log_pdf(yn)
        mix = np.log(np.sum(np.exp(log_pdf) * model.weight * mask, axis=0))
        ll = mix.sum()
        assert ll >= prev - 1e-6 * abs(prev if np.isfinite(prev) else 1.0)
        prev = ll
    post = model.predict(y)
    assert np.max(np.abs(post.sum(axis=0) - 1)) < 1e-12
    # invariance to per-frame complex scaling
    s = np.exp(rng.standard_normal((T, 1))) * np.exp(1j * rng.uniform(0, 6, (T, 1)))
    post2 = trainer.fit(y * s, init, iterations=7, source_activity_mask=mask).predict(y * s)
    assert np.max(np.abs(post - post2)) < 1e-9
    # eigenvalues are max-normalised and floored
    assert np.all(model.covariance_eigenvalues <= 1 + 1e-15)
    assert np.all(model.covariance_eigenvalues >= 1e-10)


def test_gss_batched_equals_per_frequency_loop():
    rng = np.random.default_rng(3)
    Obs = crandn(rng, 4, 60, 5)
    act = np.ones((3, 60), bool)
    act[0, :20] = False
    a = oracle.gss_block(Obs, act, 4, 1)
    b = oracle.gss_block_batched(Obs, act, 4, 1)
    assert a.shape == (3, 60, 5) and np.max(np.abs(a - b)) < 1e-12


def test_mvdr_distortionless_ban_and_reference_channel():
    rng = np.random.default_rng(4)
    F, D = 6, 5
    a = crandn(rng, F, D)
    sigma = 2.0
    phi_x = sigma * np.einsum('fd,fe->fde', a, a.conj())
    n = crandn(rng, F, D, 3 * D)
    phi_n = np.einsum('fdt,fet->fde', n, n.conj()) / (3 * D)
    w, ref = oracle.get_mvdr_vector_souden(phi_x, phi_n, eps=1e-10, return_ref_channel=True)
    # rank-1 target: w ~ Phi_N^-1 a and w^H a = a_ref (distortionless at ref before BAN)
    for f in range(F):
        assert abs(np.vdot(w[f], a[f]) - a[f, ref]) < 1e-9
        ideal = np.linalg.solve(phi_n[f], a[f])
        assert abs(abs(np.vdot(ideal, w[f])) - np.linalg.norm(ideal) * np.linalg.norm(w[f])) < 1e-9
    wb = oracle.blind_analytic_normalization(w, phi_n)
    for f in range(F):
        num = np.sqrt(np.abs(w[f].conj() @ phi_n[f] @ phi_n[f] @ w[f]))
        den = np.abs(w[f].conj() @ phi_n[f] @ w[f])
        assert rel_err(wb[f], w[f] * num / den) < 1e-12
    # PSD matrices are Hermitian and use the 1e-10-floored mask normalisation
    Y = crandn(rng, F, D, 40)
    m = rng.uniform(size=(F, 40))
    psd = oracle.get_power_spectral_density_matrix(Y, m)
    assert rel_err(psd, psd.conj().transpose(0, 2, 1)) < 1e-14
    assert not oracle.get_power_spectral_density_matrix(Y, np.zeros((F, 40))).any()
    # constructed reference-channel case: with Phi_X = diag(x), Phi_N = I the SNR of
    # reference channel r is x_r (a rank-1 Phi_X would tie all channels)
    phi_x = np.broadcast_to(np.diag([1.0, 1.0, 100.0, 1.0, 1.0]).astype(complex), (F, D, D)).copy()
    phi_n = np.broadcast_to(np.eye(D, dtype=complex), (F, D, D)).copy()
    _, ref = oracle.get_mvdr_vector_souden(phi_x, phi_n, eps=1e-10, return_ref_channel=True)
    assert ref == 2


def test_end_to_end_target_is_enhanced():
    """Two spatially separated sources + noise: the beamformed output correlates
    with the target's image far more than with the interferer's."""
    from pb_chime5_amd import synthetic
    u = synthetic.tiny(seed=3, num_channels=6, num_samples=20000, num_speakers=2, context=1024)
    x_hat = oracle.enhance_observation(u.obs, u.activity_array, u.target_index, u.ex, wpe=False,
                                       bss_iterations=10, gss_fn=oracle.gss_block_batched)
    assert x_hat.shape[0] >= 20000 and np.all(np.isfinite(x_hat))
    tgt = u.activity['P01'] & ~u.activity['P02']
    itf = u.activity['P02'] & ~u.activity['P01']
    if tgt.sum() > 2000 and itf.sum() > 2000:
        gain_t = np.std(x_hat[:20000][tgt]) / np.std(u.obs[0][tgt])
        gain_i = np.std(x_hat[:20000][itf]) / np.std(u.obs[0][itf])
        assert gain_t > 1.5 * gain_i


def test_wpe_psd_context_is_a_moving_average_over_existing_frames():
    """nara_wpe.wpe.get_power(psd_context=p): np.correlate with ones(2p + 1) in 'full' mode cropped
    to the centred lags, divided by the same correlation of ones -- i.e. the mean over the
    frames of [t - p, t + p] that exist.  p = 0 is the plain frame power."""
    rng = np.random.default_rng(3)
    X = rng.standard_normal((5, 37)) + 1j * rng.standard_normal((5, 37))
    raw = np.mean(np.abs(X) ** 2, axis=0)
    assert np.allclose(oracle.get_power(X, 0), raw, rtol=1e-14, atol=0)
    for p in (1, 2, 18, 40):
        got = oracle.get_power(X, p)
        assert got.shape == raw.shape
        for t in range(raw.size):
            lo, hi = max(0, t - p), min(raw.size - 1, t + p)
            assert abs(got[t] - raw[lo:hi + 1].mean()) <= 1e-13 * raw.max(), (p, t)
        inv = oracle.get_power_inverse(X, p)
        assert np.allclose(inv, 1 / np.maximum(got, 1e-10 * got.max()))
    # the context changes the WPE result, zero context reproduces the default
    Y = rng.standard_normal((3, 80)) + 1j * rng.standard_normal((3, 80))
    assert np.array_equal(oracle.wpe_v6(Y, 3, 2, 2, 0), oracle.wpe_v6(Y, 3, 2, 2))
    assert np.max(np.abs(oracle.wpe_v6(Y, 3, 2, 2, 2) - oracle.wpe_v6(Y, 3, 2, 2))) > 1e-6


def test_config3_item_lengths_are_seeded():
    """BASELINE configs[2]: the core lengths of the 512 dev-shaped items (what bench.py's sharded
    session uses as costs and cuts its items to) come from the items' own seeds."""
    from pb_chime5_amd import synthetic
    cores = [synthetic.config3_core_samples(i) for i in range(512)]
    assert cores[:3] == [synthetic.config3_core_samples(i) for i in range(3)]
    assert min(cores) >= 8000 and max(cores) <= 240000
    assert 30000 < np.median(cores) < 50000                     # median 2.5 s
    frames = [(c + 480000 + 2 * 768 - 1024 + 255) // 256 + 1 for c in cores]
    assert min(frames) >= 1906 and max(frames) <= 2816
    assert len(set(cores)) > 500


def test_normalize_observation_is_maximum_of_norm_and_tiny():
    """upstream: y / maximum(norm, tiny) -- zero frames stay zero, a frame of denormal norm is
    divided by tiny (and does NOT come out with unit length), NaN stays NaN."""
    tiny = np.finfo(np.float64).tiny
    y = np.zeros((5, 3), complex)
    y[1] = [3.0, 4.0j, 0.0]
    y[2] = [3e-310, 4e-310j, 0.0]                       # norm 5e-310 < tiny
    y[3] = [np.nan, 1.0, 0.0]
    y[4] = [tiny, 0.0, 0.0]                             # norm == tiny: unit length
    yn = oracle.normalize_observation(y)                # (D, T)
    assert yn.shape == (3, 5)
    assert np.all(yn[:, 0] == 0)
    assert np.allclose(yn[:, 1], [0.6, 0.8j, 0.0])
    assert np.array_equal(yn[:, 2], y[2] / tiny)
    assert abs(np.linalg.norm(yn[:, 2]) - 5e-310 / tiny) < 1e-12
    assert np.isnan(yn[0, 3])
    assert yn[0, 4] == 1.0
