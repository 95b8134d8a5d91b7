"""CPU oracle for the pb_chime5 WPE -> CACGMM-GSS -> MVDR hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package (``pb_chime5_amd``)
may import this module; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` use it, and only as the checker / the
reported CPU baseline -- never as the thing shipped or measured as "value".

What it is
----------
A float64 / complex128 NumPy restatement of the arithmetic that
``/root/reference/pb_chime5/core.py:514-571`` (``Enhancer.enhance_observation``)
executes per utterance.  The reference keeps none of that arithmetic in-tree: it
calls third-party packages that are absent from ``/root/reference`` and from this
image (no network):

* ``nara_wpe`` (``setup.py:142`` ``nara_wpe>=0.0.6``) -- ``utils.stft`` /
  ``utils.istft`` / ``utils._samples_to_stft_frames`` / ``wpe.wpe_v8``
  (call sites ``core.py:52,224,306,315``);
* ``pb_bss`` (git submodule ``.gitmodules:1-3``, empty directory, commit
  unknown) -- ``distribution.CACGMMTrainer`` (``core.py:165,180,195``) and
  ``extraction.beamformer.*`` (``speech_enhancement/beamforming_wrapper.py:51-97``).

Each function below restates the published algorithm of the named upstream
function and cites the reference call site it serves.

Parity status
-------------
* PINNED by reference known-answer vectors (tests/test_oracle_golden.py):
  ``stft`` framing / fading / padding / rfft scaling (doctest
  ``database/chime5/database.py:417-453``), ``activity_time_to_frequency``
  (same doctest), ``segment_axis`` (``utils/numpy_utils.py:42-136``),
  ``stable_solve`` (``math/solve.py:38-87``), and -- through fixtures captured
  by running the reference's own ``core.py`` / ``beamforming_wrapper.py``
  orchestration in the build container (``tests/golden/make_golden.py``) --
  all activity / indexing / mask post-processing / layout logic.
* PARITY UNPINNED (no golden vector exists anywhere in the reference): the
  Blackman window choice, ``istft``, ``wpe_v8``, ``CACGMMTrainer.fit/predict``,
  PSD / MVDR-Souden / BAN / apply.  For those the restatement follows the
  upstream algorithm as recalled in SURVEY.md section 8a / appendix A and is
  checked by invariants only (tests/test_oracle_invariants.py).
"""
import math

import numpy as np

TINY = np.finfo(np.float64).tiny


# --------------------------------------------------------------------------
# framing helpers
# --------------------------------------------------------------------------
def segment_axis(x, length, shift, end='pad'):
    """Last-axis framing, restating ``segment_axis_v2`` semantics
    (/root/reference/pb_chime5/utils/numpy_utils.py:10-222; the in-tree twin of
    the helper nara_wpe's stft uses).  ``end`` in {'pad', 'cut', None}."""
    x = np.asarray(x)
    n = x.shape[-1]
    if end == 'pad':
        if n < length:
            pad = length - n
        elif shift != 1 and (n + shift - length) % shift != 0:
            pad = shift - ((n + shift - length) % shift)
        else:
            pad = 0
        if pad:
            x = np.concatenate(
                [x, np.zeros(x.shape[:-1] + (pad,), dtype=x.dtype)], axis=-1)
            n = x.shape[-1]
    elif end is None:
        assert (n + shift - length) % shift == 0, (n, shift, length)
    elif end != 'cut':
        raise ValueError(end)
    num = (n + shift - length) // shift
    num = max(num, 0)
    idx = np.arange(num)[:, None] * shift + np.arange(length)[None, :]
    return x[..., idx]


def stft_frames(num_samples, size=1024, shift=256, fading=True):
    """Number of STFT frames for ``pad=True`` (SURVEY.md section 8 header)."""
    n = num_samples + (2 * (size - shift) if fading else 0)
    if n < size:
        return 1
    return -(-(n - size) // shift) + 1


def samples_to_stft_frames(samples, size, shift, *, pad=True, fading=False):
    """``nara_wpe.utils._samples_to_stft_frames`` (call site
    /root/reference/pb_chime5/core.py:224-237)."""
    if fading:
        samples = samples + 2 * (size - shift)
    frames = (samples - size + shift) / shift
    if pad:
        return math.ceil(frames)
    return math.floor(frames)


def start_end_context_frames(ex, stft_size, stft_shift, stft_fading):
    """/root/reference/pb_chime5/core.py:217-238."""
    start_context_samples = ex['start_orig']['original'] - ex['start']['original']
    end_context_samples = ex['end']['original'] - ex['end_orig']['original']
    assert start_context_samples >= 0, (start_context_samples, ex)
    assert end_context_samples >= 0, (end_context_samples, ex)
    return (
        samples_to_stft_frames(start_context_samples, stft_size, stft_shift,
                               fading=stft_fading),
        samples_to_stft_frames(end_context_samples, stft_size, stft_shift,
                               fading=stft_fading),
    )


def activity_time_to_frequency(time_activity, stft_window_length, stft_shift,
                               stft_fading, stft_pad=True):
    """/root/reference/pb_chime5/database/chime5/database.py:409-472."""
    time_activity = np.asarray(time_activity)
    if stft_fading:
        pad = stft_window_length - stft_shift
        z = np.zeros(time_activity.shape[:-1] + (pad,), dtype=time_activity.dtype)
        time_activity = np.concatenate([z, time_activity, z], axis=-1)
    return segment_axis(
        time_activity, stft_window_length, stft_shift,
        end='pad' if stft_pad else 'cut').any(axis=-1)


# --------------------------------------------------------------------------
# STFT / iSTFT  (nara_wpe.utils.stft / istft; core.py:305-321)
# --------------------------------------------------------------------------
def blackman_periodic(size):
    """``scipy.signal.blackman(size + 1)[:-1]`` -- the ``symmetric_window=False``
    branch of nara_wpe's stft (window choice: parity unpinned)."""
    from scipy.signal.windows import blackman
    return blackman(size + 1)[:-1]


def stft(time_signal, size=1024, shift=256, window=None, fading=True, pad=True):
    """(..., N) real -> (..., T, size//2+1) complex128."""
    time_signal = np.asarray(time_signal, dtype=np.float64)
    if fading:
        p = size - shift
        z = np.zeros(time_signal.shape[:-1] + (p,))
        time_signal = np.concatenate([z, time_signal, z], axis=-1)
    if window is None:
        w = blackman_periodic(size)
    elif callable(window):
        w = np.asarray(window(size + 1)[:-1], dtype=np.float64)
    else:
        w = np.asarray(window, dtype=np.float64)
    seg = segment_axis(time_signal, size, shift, end='pad' if pad else 'cut')
    return np.fft.rfft(seg * w, n=size, axis=-1)


def biorthogonal_window(analysis_window, shift):
    """Synthesis window of nara_wpe's istft (``_biorthogonal_window_loopy``;
    SURVEY.md appendix A): analysis / sum over shifts of analysis**2, where the
    upstream loop leaves out the very last sample (``analysis_index + 1 <
    fft_size``).  The upstream ``/ fft_size`` is undone by its ``window *= size``
    and is therefore omitted here."""
    w = np.asarray(analysis_window, dtype=np.float64)
    size = len(w)
    assert size % shift == 0
    number_of_shifts = size // shift
    sum_of_squares = np.zeros(shift)
    for synthesis_index in range(shift):
        for sample_index in range(number_of_shifts + 1):
            analysis_index = synthesis_index + sample_index * shift
            if analysis_index + 1 < size:
                sum_of_squares[synthesis_index] += w[analysis_index] ** 2
    sum_of_squares = np.kron(np.ones(number_of_shifts), sum_of_squares)
    return w / sum_of_squares


def istft(stft_signal, size=1024, shift=256, window=None, fading=True):
    """(..., T, size//2+1) complex -> (..., N') float64; overlap-add in
    increasing frame order (np.add.at upstream)."""
    stft_signal = np.asarray(stft_signal)
    assert stft_signal.shape[-1] == size // 2 + 1, stft_signal.shape
    if window is None:
        w = blackman_periodic(size)
    else:
        w = np.asarray(window, dtype=np.float64)
    syn = biorthogonal_window(w, shift)
    T = stft_signal.shape[-2]
    frames = syn * np.fft.irfft(stft_signal, n=size, axis=-1)[..., :size]
    out = np.zeros(stft_signal.shape[:-2] + (T * shift + size - shift,))
    for t in range(T):
        out[..., t * shift:t * shift + size] += frames[..., t, :]
    if fading:
        out = out[..., size - shift:out.shape[-1] - (size - shift)]
    return out


# --------------------------------------------------------------------------
# linear solve with lstsq fallback (pb_chime5/math/solve.py:20-114)
# --------------------------------------------------------------------------
def stable_solve(A, B):
    A = np.asarray(A)
    B = np.asarray(B)
    assert A.shape[:-2] == B.shape[:-2], (A.shape, B.shape)
    assert A.shape[-1] == B.shape[-2], (A.shape, B.shape)
    try:
        return np.linalg.solve(A, B)
    except np.linalg.LinAlgError:
        shape_B = B.shape
        A2 = A.reshape((-1,) + A.shape[-2:])
        B2 = B.reshape((-1,) + B.shape[-2:])
        C = np.zeros_like(B2)
        for i in range(A2.shape[0]):
            try:
                C[i] = np.linalg.solve(A2[i], B2[i])
            except np.linalg.LinAlgError:
                C[i] = np.linalg.lstsq(A2[i], B2[i], rcond=None)[0]
        return C.reshape(shape_B)


# --------------------------------------------------------------------------
# WPE  (nara_wpe.wpe.wpe_v8 -> wpe_v6; core.py:48-88)
# --------------------------------------------------------------------------
def build_y_tilde(Y, taps, delay):
    """Y (D, T) -> Y_tilde (taps*D, T); block j (0..taps-1) holds Y delayed by
    ``delay + taps - 1 - j`` frames (largest delay first, as upstream), zeros
    where the index is negative."""
    D, T = Y.shape
    out = np.zeros((taps, D, T), dtype=Y.dtype)
    for j in range(taps):
        d = delay + taps - 1 - j
        if d < T:
            out[j, :, d:] = Y[:, :T - d]
    return out.reshape(taps * D, T)


def get_power(signal, psd_context=0):
    """nara_wpe.wpe.get_power: mean_d |X|^2 per frame; psd_context = p > 0 (an int;
    core.py:583 exposes it as wpe_psd_context, default 0) averages it over the frames
    t-p..t+p that exist: np.correlate with ones(2p + 1) in 'full' mode, cropped to the T
    centred lags and divided by the same correlation of an all-ones signal."""
    power = np.mean(signal.real ** 2 + signal.imag ** 2, axis=-2)
    if np.isposinf(psd_context):            # upstream: the global mean for every frame
        return np.broadcast_to(np.mean(power, axis=-1, keepdims=True), power.shape).copy()
    if psd_context != 0:
        assert int(psd_context) == psd_context and psd_context > 0, psd_context
        p = int(psd_context)
        kernel = np.ones(2 * p + 1)
        power = np.correlate(power, kernel, mode='full')[p:-p]
        power = power / np.correlate(np.ones_like(power), kernel, mode='full')[p:-p]
    return power


def get_power_inverse(signal, psd_context=0):
    """1 / max(power, 1e-10 * max_t power)."""
    power = get_power(signal, psd_context)
    eps = 1e-10 * np.max(power)
    return 1 / np.maximum(power, eps)


def wpe_v6(Y, taps=10, delay=3, iterations=3, psd_context=0):
    """One frequency: Y (D, T) complex128 -> X (D, T); statistics_mode='full'."""
    X = np.copy(Y)
    Y_tilde = build_y_tilde(Y, taps, delay)
    for _ in range(iterations):
        inverse_power = get_power_inverse(X, psd_context=psd_context)
        Y_tilde_inverse_power = Y_tilde * inverse_power[None, :]
        R = Y_tilde_inverse_power @ Y_tilde.conj().T
        P = Y_tilde_inverse_power @ Y.conj().T
        G = stable_solve(R, P)
        X = Y - G.conj().T @ Y_tilde
    return X


def wpe_v8(Y, taps=10, delay=3, iterations=3, psd_context=0):
    """(..., D, T): loops ``wpe_v6`` over the leading (frequency) axes."""
    Y = np.asarray(Y)
    if Y.ndim == 2:
        return wpe_v6(Y, taps, delay, iterations, psd_context)
    out = np.empty_like(Y)
    for index in np.ndindex(Y.shape[:-2]):
        out[index] = wpe_v6(Y[index], taps, delay, iterations, psd_context)
    return out


def wpe_block(Obs, taps=10, delay=2, iterations=3, psd_context=0):
    """``WPE.__call__`` for 3-D input (core.py:50-58): Obs (D,T,F)->(D,T,F)."""
    return wpe_v8(Obs.transpose(2, 0, 1), taps=taps, delay=delay,
                  iterations=iterations, psd_context=psd_context
                  ).transpose(1, 2, 0)


# --------------------------------------------------------------------------
# CACGMM  (pb_bss.distribution.CACGMMTrainer / CACGMM; core.py:154-214)
# --------------------------------------------------------------------------
def normalize_observation(y):
    """(..., T, D) -> (..., D, T), unit norm per frame: y / maximum(norm, tiny), as upstream's
    ``normalize_observation`` writes it (a frame whose norm is a denormal number is divided by
    tiny, not by its norm; NaN stays NaN; with sqrt(sum |y|^2) a norm in (0, tiny) cannot
    occur at all -- the squares underflow first -- so this equals the 'where' eps style on every
    float64 input)."""
    norm = np.linalg.norm(y, axis=-1, keepdims=True)
    norm = np.maximum(norm, TINY)
    return np.ascontiguousarray(np.swapaxes(y / norm, -2, -1))


class CACGMM:
    """weight (..., K, 1); eigenvectors (..., K, D, D); eigenvalues (..., K, D)
    (already max-normalised and floored)."""

    def __init__(self, weight, eigenvectors, eigenvalues):
        self.weight = weight
        self.covariance_eigenvectors = eigenvectors
        self.covariance_eigenvalues = eigenvalues

    @property
    def log_determinant(self):
        return np.sum(np.log(self.covariance_eigenvalues), axis=-1)

    def _inverse_covariance(self):
        V = self.covariance_eigenvectors
        return (V / self.covariance_eigenvalues[..., None, :]) @ \
            np.swapaxes(V.conj(), -1, -2)

    def _log_pdf(self, y):
        """y (..., D, T) normalised -> log_pdf (..., K, T), quadratic (..., K, T)."""
        D = y.shape[-2]
        Minv = self._inverse_covariance()
        quadratic_form = np.maximum(
            np.abs(np.einsum('...dt,...kde,...et->...kt', y.conj(), Minv, y,
                             optimize=True)),
            TINY)
        log_pdf = -D * np.log(quadratic_form)
        log_pdf -= self.log_determinant[..., None]
        return log_pdf, quadratic_form

    def _predict(self, y, source_activity_mask=None, affiliation_eps=0.):
        log_pdf, quadratic_form = self._log_pdf(y)
        affiliation = log_pdf - np.amax(log_pdf, axis=-2, keepdims=True)
        np.exp(affiliation, out=affiliation)
        affiliation *= self.weight
        if source_activity_mask is not None:
            assert source_activity_mask.dtype == bool, source_activity_mask.dtype
            affiliation *= source_activity_mask
        denominator = np.maximum(
            np.sum(affiliation, axis=-2, keepdims=True), TINY)
        affiliation /= denominator
        if affiliation_eps != 0:
            affiliation = np.clip(affiliation, affiliation_eps,
                                  1 - affiliation_eps)
        return affiliation, quadratic_form

    def predict(self, y, source_activity_mask=None):
        """y (..., T, D) -> affiliation (..., K, T); affiliation_eps = 0."""
        assert np.iscomplexobj(y), y.dtype
        y = normalize_observation(y)
        affiliation, _ = self._predict(
            y, source_activity_mask=source_activity_mask)
        return affiliation


class CACGMMTrainer:
    def fit(self, y, initialization, iterations=100, *,
            source_activity_mask=None, affiliation_eps=1e-10,
            eigenvalue_floor=1e-10):
        """y (..., T, D) complex; initialization (..., K, T) array or CACGMM.
        ``iterations`` M-steps, ``iterations - 1`` E-steps when initialised from
        an affiliation array."""
        assert np.iscomplexobj(y), y.dtype
        assert y.shape[-1] > 1, y.shape
        y = normalize_observation(y)
        D, T = y.shape[-2:]
        if isinstance(initialization, CACGMM):
            model = initialization
            affiliation = quadratic_form = None
        else:
            affiliation = np.asarray(initialization, dtype=np.float64)
            quadratic_form = np.ones(affiliation.shape, dtype=np.float64)
            model = None
            assert affiliation.shape[-2] < 20 and D < 35
        for _ in range(iterations):
            if model is not None:
                affiliation, quadratic_form = model._predict(
                    y, source_activity_mask=source_activity_mask,
                    affiliation_eps=affiliation_eps)
            model = self._m_step(y, quadratic_form, affiliation,
                                 eigenvalue_floor)
        return model

    @staticmethod
    def _m_step(y, quadratic_form, affiliation, eigenvalue_floor):
        D = y.shape[-2]
        weight = np.mean(affiliation, axis=-1, keepdims=True)
        denominator = np.maximum(np.sum(affiliation, axis=-1), TINY)
        quadratic_form = np.maximum(quadratic_form, 10 * TINY)
        covariance = D * np.einsum(
            '...dt,...et,...kt->...kde', y, y.conj(),
            affiliation / quadratic_form, optimize=True)
        covariance /= denominator[..., None, None]
        covariance = (covariance + np.swapaxes(covariance.conj(), -1, -2)) / 2
        eigenvals, eigenvecs = np.linalg.eigh(covariance)
        eigenvals = eigenvals / np.maximum(
            np.amax(eigenvals, axis=-1, keepdims=True), TINY)
        eigenvals = np.maximum(eigenvals, eigenvalue_floor)
        return CACGMM(weight, eigenvecs, eigenvals)


def gss_initialization(activity_freq):
    """core.py:156-163 (without the hard-coded 513-fold repeat)."""
    init = np.asarray(activity_freq, dtype=np.float64)
    init = np.where(init == 0, 1e-10, init)
    init = init / np.sum(init, keepdims=True, axis=0)
    return init, np.asarray(activity_freq, dtype=bool)


def gss_block(Obs, activity_freq, iterations=20, iterations_post=1,
              frequencies=None):
    """``GSS.__call__`` (core.py:154-214): Obs (D,T,F), activity (K,T) bool ->
    posterior (K,T,F).  Per-frequency Python loop like the reference."""
    init, mask = gss_initialization(activity_freq)
    trainer = CACGMMTrainer()
    F = Obs.shape[-1]
    T = Obs.shape[-2]
    fs = range(F) if frequencies is None else frequencies
    out = []
    for f in fs:
        y = Obs.T[f]
        cur = trainer.fit(y, init[..., :T], iterations=iterations,
                          source_activity_mask=mask[..., :T])
        if iterations_post != 0:
            if iterations_post != 1:
                cur = trainer.fit(y, cur, iterations=iterations_post - 1)
            aff = cur.predict(y)
        else:
            aff = cur.predict(y, source_activity_mask=mask[..., :T])
        out.append(aff)
    return np.array(out).transpose(1, 2, 0)


def gss_block_batched(Obs, activity_freq, iterations=20, iterations_post=1):
    """Same arithmetic as ``gss_block`` with the frequency loop folded into the
    leading batch axis (used by tests at sizes where the loop is too slow)."""
    init, mask = gss_initialization(activity_freq)
    trainer = CACGMMTrainer()
    y = np.ascontiguousarray(Obs.transpose(2, 1, 0))  # (F, T, D)
    cur = trainer.fit(y, init[None], iterations=iterations,
                      source_activity_mask=mask[None])
    if iterations_post != 0:
        if iterations_post != 1:
            cur = trainer.fit(y, cur, iterations=iterations_post - 1)
        aff = cur.predict(y)
    else:
        aff = cur.predict(y, source_activity_mask=mask[None])
    return aff.transpose(1, 2, 0)


# --------------------------------------------------------------------------
# beamforming (pb_bss.extraction.beamformer; beamforming_wrapper.py:11-124)
# --------------------------------------------------------------------------
def get_power_spectral_density_matrix(observation, mask, normalize=True):
    """observation (F, D, T), mask (F, T) -> (F, D, D)."""
    mask = np.array(mask, dtype=np.float64)
    if normalize:
        mask = mask / np.maximum(np.sum(mask, axis=-1, keepdims=True), 1e-10)
    return np.einsum('...dt,...et->...de', mask[..., None, :] * observation,
                     observation.conj())


def get_optimal_reference_channel(w_mat, target_psd_matrix, noise_psd_matrix,
                                  eps=None):
    if w_mat.ndim != 3:
        raise ValueError(w_mat.shape)
    if eps is None:
        eps = TINY
    num = np.einsum('FdR,FdD,FDR->R', w_mat.conj(), target_psd_matrix, w_mat)
    den = np.einsum('FdR,FdD,FDR->R', w_mat.conj(), noise_psd_matrix, w_mat)
    SNR = num / np.maximum(den, eps)
    assert np.all(np.isfinite(SNR)), SNR
    return int(np.argmax(SNR.real))


def get_mvdr_vector_souden(target_psd_matrix, noise_psd_matrix,
                           ref_channel=None, eps=None,
                           return_ref_channel=False):
    phi = stable_solve(noise_psd_matrix, target_psd_matrix)
    lambda_ = np.trace(phi, axis1=-1, axis2=-2)[..., None, None]
    if eps is None:
        eps = TINY
    mat = phi / np.maximum(lambda_.real, eps)
    if ref_channel is None:
        ref_channel = get_optimal_reference_channel(
            mat, target_psd_matrix, noise_psd_matrix, eps=eps)
    beamformer = mat[..., ref_channel]
    if return_ref_channel:
        return beamformer, ref_channel
    return beamformer


def blind_analytic_normalization(vector, noise_psd_matrix, eps=0):
    nominator = np.einsum('...a,...ab,...bc,...c->...', vector.conj(),
                          noise_psd_matrix, noise_psd_matrix, vector)
    nominator = np.abs(np.sqrt(nominator))
    denominator = np.einsum('...a,...ab,...b->...', vector.conj(),
                            noise_psd_matrix, vector)
    denominator = np.abs(denominator)
    with np.errstate(invalid='ignore', divide='ignore'):
        normalization = nominator / (denominator + eps)
    return vector * normalization[..., None]


def get_gev_vector(target_psd_matrix, noise_psd_matrix):
    """pb_bss get_gev_vector (call site beamforming_wrapper.py:79): principal
    generalised eigenvector of (Phi_X, Phi_N) per frequency, B-normalised like
    scipy.linalg.eigh / Eigen's GeneralizedSelfAdjointEigenSolver; phase arbitrary."""
    from scipy.linalg import eigh
    F, D, _ = target_psd_matrix.shape
    out = np.empty((F, D), dtype=np.complex128)
    for f in range(F):
        _, vecs = eigh(target_psd_matrix[f], noise_psd_matrix[f])
        out[f] = vecs[:, -1]
    return out


def beamform_gev_from_masks(Y, X_mask, N_mask, ban=True):
    """beamforming_wrapper.py:192-208 for Y (D,T,F) and 2-D masks (T,F).  (The wrapper builds
    the same _Beamformer object as the MVDR entry point: its ``assert D < 30``,
    beamforming_wrapper.py:44, applies here too.)"""
    Yf = Y.transpose(2, 0, 1)
    F, D, T = Yf.shape
    assert D < 30, (D, Yf.shape)
    cov_x = get_power_spectral_density_matrix(Yf, X_mask.T)
    cov_n = get_power_spectral_density_matrix(Yf, N_mask.T)
    w = get_gev_vector(cov_x, cov_n)
    if ban:
        w = blind_analytic_normalization(w, cov_n)
    return apply_beamforming_vector(w, Yf).T


def apply_beamforming_vector(vector, mix):
    """vector (F, D), mix (F, D, T) -> (F, T)."""
    return np.einsum('...a,...at->...t', vector.conj(), mix)


def beamform_mvdr_souden_from_masks(Y, X_mask, N_mask, ban=False,
                                    return_details=False):
    """beamforming_wrapper.py:108-124 for Y (D,T,F) and 2-D masks (T,F)."""
    Yf = Y.transpose(2, 0, 1)
    Xm = X_mask.T
    Nm = N_mask.T
    F, D, T = Yf.shape
    assert D < 30, (D, Yf.shape)
    assert Xm.shape == (F, T) and Nm.shape == (F, T)
    cov_x = get_power_spectral_density_matrix(Yf, Xm)
    cov_n = get_power_spectral_density_matrix(Yf, Nm)
    w, ref = get_mvdr_vector_souden(cov_x, cov_n, eps=1e-10,
                                    return_ref_channel=True)
    if ban:
        w = blind_analytic_normalization(w, cov_n)
    X_hat = apply_beamforming_vector(w, Yf).T
    if return_details:
        return X_hat, dict(cov_x=cov_x, cov_n=cov_n, w=w, ref_channel=ref)
    return X_hat


# --------------------------------------------------------------------------
# per-utterance pipeline (core.py:514-571)
# --------------------------------------------------------------------------
def enhance_observation(obs, activity, target_index, ex=None, *,
                        wpe=True, wpe_taps=10, wpe_delay=2, wpe_iterations=3,
                        wpe_psd_context=0,
                        stft_size=1024, stft_shift=256, stft_fading=True,
                        bss_iterations=20, bss_iterations_post=1,
                        bf_drop_context=True, bf='mvdrSouden_ban',
                        postfilter=None, return_details=False,
                        gss_fn=None, wpe_fn=None):
    """obs (D,N) float64; activity (K,N) bool in dict order; target_index = the
    row of the target speaker.  Returns x_hat (N',) float64."""
    Obs = stft(obs, stft_size, stft_shift, fading=stft_fading)
    if wpe:
        Obs = (wpe_block if wpe_fn is None else wpe_fn)(
            Obs, wpe_taps, wpe_delay, wpe_iterations, wpe_psd_context)
    activity_freq = activity_time_to_frequency(
        np.asarray(activity), stft_size, stft_shift, stft_fading, stft_pad=True)
    gss = gss_block if gss_fn is None else gss_fn
    masks = gss(Obs, activity_freq, iterations=bss_iterations,
                iterations_post=bss_iterations_post)
    start_frames = end_frames = 0
    if bf_drop_context:
        start_frames, end_frames = start_end_context_frames(
            ex, stft_size, stft_shift, stft_fading)
        masks[:, :start_frames, :] = 0
        if end_frames > 0:
            masks[:, -end_frames:, :] = 0
    target_mask = masks[target_index]
    distortion_mask = np.sum(np.delete(masks, target_index, axis=0), axis=0)
    details = {}
    if bf == 'mvdrSouden_ban':
        X_hat, details = beamform_mvdr_souden_from_masks(
            Obs, target_mask, distortion_mask, ban=True, return_details=True)
    elif bf == 'gev_ban':
        X_hat = beamform_gev_from_masks(Obs, target_mask, distortion_mask, ban=True)
    elif bf == 'ch2':
        X_hat = Obs[2]
    elif bf == 'sum':
        X_hat = np.sum(Obs, axis=0)
    else:
        raise NotImplementedError(bf)
    if postfilter is None:
        pass
    elif postfilter == 'mask_mul':
        X_hat = X_hat * target_mask
    else:
        raise NotImplementedError(postfilter)
    x_hat = istft(X_hat, stft_size, stft_shift, fading=stft_fading)
    if return_details:
        details.update(Obs=Obs, activity_freq=activity_freq, masks=masks,
                       target_mask=target_mask, distortion_mask=distortion_mask,
                       X_hat=X_hat, start_context_frames=start_frames,
                       end_context_frames=end_frames)
        return x_hat, details
    return x_hat
