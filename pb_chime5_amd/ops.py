"""NumPy-in / NumPy-out operators of the hot path, executed on the GPU through the
C ABI (include/gss_hip.h).  Shapes, dtypes and argument meaning follow the
third-party functions the reference calls, so the blocks in ``core.py`` read like
the reference's:

=========================  ====================================================
here                        reference call (file:line)
=========================  ====================================================
``stft`` / ``istft``        nara_wpe.utils.stft / istft   (core.py:305-321)
``wpe_v8``                  nara_wpe.wpe.wpe_v8           (core.py:52-58)
``cacgmm_posteriors``       CACGMMTrainer.fit + predict   (core.py:165-208)
``mvdr_souden_from_masks``  pb_bss beamformer chain       (beamforming_wrapper.py:51-97)
``enhance_observation``     Enhancer.enhance_observation  (core.py:514-571)
=========================  ====================================================

All of them raise if libgss_hip.so or a GPU is missing.
"""
import ctypes

import numpy as np

from . import _capi
from ._capi import Context, GssDebugTaps, GssParams, c_void_p, default_context

_BF_CODES = {'mvdrSouden_ban': 0, 'ch2': 1, 'sum': 2, 'gev_ban': 3}
_POSTFILTER_CODES = {None: 0, 'mask_mul': 1}


# --------------------------------------------------------------------------
# windows (host logic, float64; nara_wpe.utils.stft / istft)
# --------------------------------------------------------------------------
def analysis_window(size, window=None):
    """Periodic window ``window(size + 1)[:-1]`` (nara_wpe ``symmetric_window=False``);
    default scipy.signal blackman."""
    if window is None:
        from scipy.signal.windows import blackman as window
    if callable(window):
        return np.asarray(window(size + 1)[:-1], dtype=np.float64)
    w = np.asarray(window, dtype=np.float64)
    assert w.shape == (size,), (w.shape, size)
    return w


def synthesis_window(analysis, shift):
    """Biorthogonal synthesis window as nara_wpe's istft builds it: analysis /
    (sum over the size/shift shifted copies of analysis**2), where the upstream
    loop leaves the very last sample out of that sum."""
    w = np.asarray(analysis, dtype=np.float64)
    size = len(w)
    assert size % shift == 0, (size, shift)
    number_of_shifts = size // shift
    sq = w ** 2
    sq[-1] = 0.0
    # upstream accumulates shift by shift in increasing order
    sum_of_squares = np.zeros(shift)
    for j in range(number_of_shifts):
        sum_of_squares += sq[j * shift:(j + 1) * shift]
    return w / np.tile(sum_of_squares, number_of_shifts)


def _prepare_windows(ctx, size, shift, window=None):
    # (the shift has to divide the window: nara_wpe's istft asserts it for its synthesis window,
    # and the library keeps one (size, shift) pair of windows per context for both directions)
    a = analysis_window(size, window)
    s = synthesis_window(a, shift)
    ctx.set_windows(size, shift, a, s)


def stft_frames(num_samples, size, shift, fading=True):
    return int(_capi.load_library().gss_stft_num_frames(
        int(num_samples), size, shift, int(bool(fading))))


def samples_to_stft_frames(samples, size, shift, *, fading=False):
    """nara_wpe.utils._samples_to_stft_frames(pad=True) (core.py:224-237)."""
    return int(_capi.load_library().gss_samples_to_stft_frames(
        int(samples), size, shift, int(bool(fading))))


# --------------------------------------------------------------------------
# stage operators
# --------------------------------------------------------------------------
def stft(time_signal, size=1024, shift=256, *, window=None, fading=True, ctx=None):
    """(..., N) real -> (..., T, size//2+1) complex128."""
    ctx = ctx or default_context()
    x = np.asarray(time_signal, dtype=np.float64)
    lead = x.shape[:-1]
    N = x.shape[-1]
    D = int(np.prod(lead, dtype=np.int64)) if lead else 1
    x2 = np.ascontiguousarray(x.reshape(D, N))
    _prepare_windows(ctx, size, shift, window)
    F = size // 2 + 1
    T = stft_frames(N, size, shift, fading)
    if D == 0 or T == 0:
        return np.zeros(lead + (T, F), np.complex128)
    if N == 0:          # nothing but the fading pad: frames of zeros
        return np.zeros(lead + (T, F), np.complex128)
    x_d = ctx.to_device(x2)
    Y_d = ctx.empty(16 * F * T * D)
    O_d = ctx.empty(16 * F * T * D)
    lib = ctx.lib
    ctx._check(lib.gss_stft(ctx.handle, c_void_p(x_d.ptr), D, N, int(bool(fading)),
                            c_void_p(Y_d.ptr)), 'gss_stft')
    ctx._check(lib.gss_layout_ftd_to_dtf(ctx.handle, c_void_p(Y_d.ptr), F, T, D,
                                         c_void_p(O_d.ptr)), 'gss_layout_ftd_to_dtf')
    out = ctx.to_host(O_d, (D, T, F), np.complex128)
    return out.reshape(lead + (T, F))


def istft(stft_signal, size=1024, shift=256, *, window=None, fading=True, ctx=None):
    """(..., T, size//2+1) complex -> (..., N') float64."""
    ctx = ctx or default_context()
    X = np.asarray(stft_signal, dtype=np.complex128)
    F = size // 2 + 1
    assert X.shape[-1] == F, X.shape
    lead = X.shape[:-2]
    T = X.shape[-2]
    X2 = np.ascontiguousarray(X.reshape((-1, T, F)))
    _prepare_windows(ctx, size, shift, window)
    n_out = int(ctx.lib.gss_istft_num_samples(T, size, shift, int(bool(fading))))
    out = np.empty((X2.shape[0], n_out))
    x_d = ctx.empty(8 * max(n_out, 1))
    for i in range(X2.shape[0]):
        X_d = ctx.to_device(X2[i])
        ctx._check(ctx.lib.gss_istft(ctx.handle, c_void_p(X_d.ptr), T, int(bool(fading)),
                                     c_void_p(x_d.ptr)), 'gss_istft')
        out[i] = ctx.to_host(x_d, (n_out,), np.float64)
    return out.reshape(lead + (n_out,))


def _obs_to_device_ftd(ctx, Obs):
    """(D,T,F) complex host array -> device (F,T,D)."""
    Obs = np.asarray(Obs, dtype=np.complex128)
    assert Obs.ndim == 3, Obs.shape
    D, T, F = Obs.shape
    src = ctx.to_device(Obs)
    dst = ctx.empty(16 * F * T * D)
    ctx._check(ctx.lib.gss_layout_dtf_to_ftd(ctx.handle, c_void_p(src.ptr), D, T, F,
                                             c_void_p(dst.ptr)), 'gss_layout_dtf_to_ftd')
    return dst, (D, T, F)


def _ftd_to_host_dtf(ctx, buf, D, T, F):
    tmp = ctx.empty(16 * F * T * D)
    ctx._check(ctx.lib.gss_layout_ftd_to_dtf(ctx.handle, c_void_p(buf.ptr), F, T, D,
                                             c_void_p(tmp.ptr)), 'gss_layout_ftd_to_dtf')
    return ctx.to_host(tmp, (D, T, F), np.complex128)


PSD_CONTEXT_ALL = 2 ** 31 - 1      # "every frame": what np.inf means upstream


def check_psd_context(psd_context):
    """nara_wpe.wpe.get_power takes a non-negative integer (frames either side of t, default 0:
    core.py:56,583) or np.inf (the mean over all frames); both are implemented -- np.inf as a
    window that always covers the whole utterance.  Everything else ((left, right) tuples --
    an argument of upstream's window_mean, not of get_power --, fractions, negative values)
    is refused in ONE place with NotImplementedError: never truncated, never a TypeError /
    OverflowError out of int()."""
    if isinstance(psd_context, (float, np.floating)) and np.isposinf(psd_context):
        return PSD_CONTEXT_ALL
    ok = isinstance(psd_context, (int, np.integer)) and not isinstance(psd_context, bool)
    if not ok and isinstance(psd_context, (float, np.floating)):
        ok = np.isfinite(psd_context) and float(psd_context).is_integer()
    if not ok or psd_context < 0 or psd_context > PSD_CONTEXT_ALL:
        raise NotImplementedError(
            f'psd_context={psd_context!r}: a non-negative integer number of frames or np.inf')
    return int(psd_context)


def wpe_dtf(Obs, taps=10, delay=2, iterations=3, psd_context=0, *, ctx=None):
    """WPE on the reference's (D,T,F) layout (what ``WPE.__call__`` hands over
    transposed to wpe_v8 and transposes back, core.py:52-58)."""
    psd_context = check_psd_context(psd_context)
    ctx = ctx or default_context()
    Y_d, (D, T, F) = _obs_to_device_ftd(ctx, Obs)
    X_d = ctx.empty(16 * F * T * D)
    ctx._check(ctx.lib.gss_wpe(ctx.handle, c_void_p(Y_d.ptr), F, T, D, int(taps), int(delay),
                               int(iterations), int(psd_context), c_void_p(X_d.ptr)), 'gss_wpe')
    return _ftd_to_host_dtf(ctx, X_d, D, T, F)


def get_power_inverse(signal, psd_context=0, *, ctx=None):
    """nara_wpe.wpe.get_power_inverse: signal (F, D, T) -> (F, T), the weights of one WPE
    iteration (mean channel power, smoothed over [t - psd_context, t + psd_context], floored
    at 1e-10 of its maximum over time, inverted)."""
    psd_context = check_psd_context(psd_context)
    ctx = ctx or default_context()
    signal = np.asarray(signal, dtype=np.complex128)
    Y_d, (D, T, F) = _obs_to_device_ftd(ctx, signal.transpose(1, 2, 0))
    w_d = ctx.empty(8 * F * T)
    ctx._check(ctx.lib.gss_wpe_inverse_power(ctx.handle, c_void_p(Y_d.ptr), F, T, D,
                                             int(psd_context), c_void_p(w_d.ptr)),
               'gss_wpe_inverse_power')
    return ctx.to_host(w_d, (F, T), np.float64)


def wpe_v8(Y, taps=10, delay=3, iterations=3, psd_context=0, *, ctx=None):
    """nara_wpe.wpe.wpe_v8 signature: Y (..., D, T) with the frequency (independent)
    axes leading; returns the same shape."""
    Y = np.asarray(Y, dtype=np.complex128)
    if Y.ndim == 2:
        return wpe_v8(Y[None], taps, delay, iterations, psd_context, ctx=ctx)[0]
    lead = Y.shape[:-2]
    D, T = Y.shape[-2:]
    Yf = Y.reshape((-1, D, T))
    out = wpe_dtf(Yf.transpose(1, 2, 0), taps, delay, iterations, psd_context, ctx=ctx)
    return out.transpose(2, 0, 1).reshape(lead + (D, T))


def cacgmm_posteriors(Obs, activity_freq, iterations=20, iterations_post=1, *, ctx=None):
    """``GSS.__call__`` (core.py:154-214): Obs (D,T,F) complex, activity (K,T) bool
    -> posterior (K,T,F) float64."""
    ctx = ctx or default_context()
    Y_d, (D, T, F) = _obs_to_device_ftd(ctx, Obs)
    act = np.ascontiguousarray(np.asarray(activity_freq).astype(bool).astype(np.uint8))
    K = act.shape[0]
    # "T: Consider end of signal" (core.py:177-184): activity may be longer than Obs
    assert act.shape[1] >= T, (act.shape, T)
    act = np.ascontiguousarray(act[:, :T])
    act_d = ctx.to_device(act)
    g_d = ctx.empty(8 * F * K * T)
    o_d = ctx.empty(8 * F * K * T)
    ctx._check(ctx.lib.gss_cacgmm(ctx.handle, c_void_p(Y_d.ptr), F, T, D, c_void_p(act_d.ptr),
                                  K, int(iterations), int(iterations_post),
                                  c_void_p(g_d.ptr)), 'gss_cacgmm')
    # (F, K*T) -> (K*T, F)
    ctx._check(ctx.lib.gss_layout_permute_f64(ctx.handle, c_void_p(g_d.ptr), F, K * T, 1, 2,
                                              c_void_p(o_d.ptr)), 'gss_layout_permute_f64')
    return ctx.to_host(o_d, (K, T, F), np.float64)


def _mask_to_device_ft(ctx, mask, T, F):
    m = np.ascontiguousarray(np.asarray(mask, dtype=np.float64).T)   # (F,T)
    assert m.shape == (F, T), (m.shape, F, T)
    return ctx.to_device(m)


def _raise_for_ref_channel(ref, name=None):
    """The beamformer's status word (gss_last_ref_channel) as the exception the reference
    raises.  -1: pb_bss get_optimal_reference_channel ``assert np.all(np.isfinite(SNR)), SNR``;
    <= -2: pb_bss get_gev_vector re-raises scipy.linalg.eigh's LinAlgError ("Error for
    frequency f ...") when the noise PSD matrix of a frequency is not positive definite."""
    prefix = f'{name}: ' if name is not None else ''
    if ref == -1:       # (not an `assert` statement: those vanish under python -O)
        raise AssertionError(f'{prefix}get_optimal_reference_channel: the SNR is not finite')
    if -2 - (1 << 24) < ref <= -2:
        raise np.linalg.LinAlgError(f'{prefix}Error for frequency {-2 - ref}: the noise PSD '
                                    'matrix is not positive definite (get_gev_vector)')


def _check_ref_channel(ctx):
    """Raises what the reference raises for the last beamformer run on ``ctx``; returns the
    reference channel otherwise."""
    ref = ctx.last_ref_channel()
    _raise_for_ref_channel(ref)
    return ref


def mvdr_souden_from_masks(Y, X_mask, N_mask, ban=False, *, ref_channel=None,
                           return_ref_channel=False, ctx=None):
    """Y (D,T,F), 2-D masks (T,F) -> X_hat (T,F) complex128.  ``ref_channel`` names the
    reference channel (pb_bss get_mvdr_vector_souden(ref_channel=...)); None = the SNR
    argmax.  Raises AssertionError when an SNR is not finite, like the reference."""
    ctx = ctx or default_context()
    Y_d, (D, T, F) = _obs_to_device_ftd(ctx, Y)
    mx = _mask_to_device_ft(ctx, X_mask, T, F)
    mn = _mask_to_device_ft(ctx, N_mask, T, F)
    X_d = ctx.empty(16 * F * T)
    if ref_channel is None:
        ctx._check(ctx.lib.gss_mvdr_souden(ctx.handle, c_void_p(Y_d.ptr), F, T, D,
                                           c_void_p(mx.ptr), c_void_p(mn.ptr), int(bool(ban)),
                                           c_void_p(X_d.ptr), None), 'gss_mvdr_souden')
    else:
        ctx._check(ctx.lib.gss_mvdr_souden_ref(ctx.handle, c_void_p(Y_d.ptr), F, T, D,
                                               c_void_p(mx.ptr), c_void_p(mn.ptr),
                                               int(bool(ban)), int(ref_channel),
                                               c_void_p(X_d.ptr)), 'gss_mvdr_souden_ref')
    ref = _check_ref_channel(ctx)
    X_hat = ctx.to_host(X_d, (T, F), np.complex128)
    if return_ref_channel:
        return X_hat, ref
    return X_hat


def gev_from_masks(Y, X_mask, N_mask, ban=True, *, ctx=None):
    """beamform_gev_from_masks: Y (D,T,F), 2-D masks (T,F) -> X_hat (T,F).  The phase
    of a generalised eigenvector is arbitrary (upstream too); magnitudes are defined.
    Raises numpy.linalg.LinAlgError when the noise PSD matrix of a frequency is not positive
    definite, like scipy.linalg.eigh inside pb_bss get_gev_vector."""
    ctx = ctx or default_context()
    Y_d, (D, T, F) = _obs_to_device_ftd(ctx, Y)
    mx = _mask_to_device_ft(ctx, X_mask, T, F)
    mn = _mask_to_device_ft(ctx, N_mask, T, F)
    X_d = ctx.empty(16 * F * T)
    ctx._check(ctx.lib.gss_gev(ctx.handle, c_void_p(Y_d.ptr), F, T, D, c_void_p(mx.ptr),
                               c_void_p(mn.ptr), int(bool(ban)), c_void_p(X_d.ptr)), 'gss_gev')
    _check_ref_channel(ctx)
    return ctx.to_host(X_d, (T, F), np.complex128)


def activity_time_to_frequency_device(time_activity, size, shift, fading, *, ctx=None):
    """Device twin of database.chime5.activity_time_to_frequency (stft_pad=True)."""
    ctx = ctx or default_context()
    act = np.asarray(time_activity)
    lead = act.shape[:-1]
    N = act.shape[-1]
    a2 = np.ascontiguousarray((act.reshape(-1, N) != 0).astype(np.uint8))
    K = a2.shape[0]
    _prepare_windows(ctx, size, shift)
    T = stft_frames(N, size, shift, fading)
    a_d = ctx.to_device(a2)
    o_d = ctx.empty(max(K * T, 16))
    ctx._check(ctx.lib.gss_activity_time_to_frequency(
        ctx.handle, c_void_p(a_d.ptr), K, N, int(bool(fading)), c_void_p(o_d.ptr)),
        'gss_activity_time_to_frequency')
    return ctx.to_host(o_d, (K, T), np.uint8).astype(bool).reshape(lead + (T,))


# --------------------------------------------------------------------------
# fused pipeline
# --------------------------------------------------------------------------
def make_params(*, stft_size=1024, stft_shift=256, stft_fading=True, wpe=True, wpe_taps=10,
                wpe_delay=2, wpe_iterations=3, wpe_psd_context=0, bss_iterations=20,
                bss_iterations_post=1, bf_drop_context=True, bf='mvdrSouden_ban',
                postfilter=None):
    if bf not in _BF_CODES:
        raise NotImplementedError(bf)
    if postfilter not in _POSTFILTER_CODES:
        raise NotImplementedError(postfilter)
    return GssParams(
        stft_size=stft_size, stft_shift=stft_shift, stft_fading=int(bool(stft_fading)),
        wpe=int(bool(wpe)), wpe_taps=wpe_taps, wpe_delay=wpe_delay,
        wpe_iterations=wpe_iterations, bss_iterations=bss_iterations,
        bss_iterations_post=bss_iterations_post, bf_drop_context=int(bool(bf_drop_context)),
        bf=_BF_CODES[bf], postfilter=_POSTFILTER_CODES[postfilter],
        wpe_psd_context=check_psd_context(wpe_psd_context))


class ResidentUtterance:
    """An utterance whose inputs already sit in HBM (what bench.py times)."""

    def __init__(self, ctx, obs, activity, params):
        obs = np.ascontiguousarray(obs, dtype=np.float64)
        act = np.ascontiguousarray((np.asarray(activity) != 0).astype(np.uint8))
        self.ctx = ctx
        self.D, self.N = obs.shape
        self.K = act.shape[0]
        self.N_act = act.shape[1]
        self.params = params
        self.T = stft_frames(self.N, params.stft_size, params.stft_shift,
                             params.stft_fading)
        self.n_out = int(ctx.lib.gss_istft_num_samples(
            self.T, params.stft_size, params.stft_shift, params.stft_fading))
        self.obs_d = ctx.to_device(obs)
        self.act_d = ctx.to_device(act)
        self.out_d = ctx.empty(8 * max(self.n_out, 1))

    def enqueue(self, target_index, start_context, end_context, taps=None):
        ctx = self.ctx
        ctx._check(ctx.lib.gss_enhance_observation(
            ctx.handle, ctypes.byref(self.params), c_void_p(self.obs_d.ptr), self.D,
            self.N, c_void_p(self.act_d.ptr), self.K, self.N_act, int(target_index),
            int(start_context), int(end_context), c_void_p(self.out_d.ptr),
            ctypes.byref(taps) if taps is not None else None),
            'gss_enhance_observation')

    def result(self):
        x_hat = self.ctx.to_host(self.out_d, (self.n_out,), np.float64)
        if self.params.bf in (_BF_CODES['mvdrSouden_ban'], _BF_CODES['gev_ban']):
            _check_ref_channel(self.ctx)
        return x_hat


class HostStaging:
    """One utterance's inputs in page-locked host memory: ``obs`` (D, N) int16 PCM followed by
    ``act`` (K, N_act) uint8, in one block (grow-only).  Loader threads fill the rows (WAV
    slices by preadv, activity tracks by slice_into), the feeder hands the set to
    `UtterancePipeline.enqueue_staged`, which starts two DMAs from it."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.block = None
        self.obs = self.act = None
        self._retired = []

    def shape(self, D, N, K, N_act):
        off = (2 * D * N + 255) // 256 * 256
        need = off + K * N_act
        if self.block is None or self.block.nbytes < need:
            # hipHostFree waits for the whole device (every stream of every utterance in
            # flight): an outgrown block is parked until free(), and blocks at least double, so
            # a session parks a handful of them
            have = 0
            if self.block is not None:
                self._retired.append(self.block)
                have = self.block.nbytes
            self.block = self.ctx.pinned(max(int(need * 1.25) + 4096, 2 * have))
        self.obs = self.block.view((D, N), np.int16, 0)
        self.act = self.block.view((K, N_act), np.uint8, off)
        return self.obs, self.act

    def free(self):
        self.obs = self.act = None
        for block in self._retired + ([self.block] if self.block is not None else []):
            block.free()
        self.block, self._retired = None, []


class UtterancePipeline:
    """Keeps up to ``depth`` utterances in flight on one GPU, each on its own context
    (HIP stream + workspace), so that one utterance's latency-bound kernels overlap the
    other's MFMA / VALU-bound ones and host work (audio I/O, H2D, D2H, writing) overlaps
    device work -- SURVEY.md section 8e "keep >= 2 utterances in flight per GPU".

        pipe = UtterancePipeline(params, depth=2)
        for ex in examples:
            prepared = load(ex)                       # host work first ...
            if pipe.full():
                tag, x_hat = pipe.pop()               # ... then wait for the oldest
            pipe.enqueue(ex, *prepared)
        while len(pipe): tag, x_hat = pipe.pop()

    Device buffers are per slot and grow-only (no hipMalloc / hipFree between
    utterances: hipFree synchronises the whole device)."""

    def __init__(self, params, depth=2, device_id=None, window=None, first_ctx=None,
                 staging_sets=0):
        from collections import deque
        import queue
        assert depth >= 1
        first = first_ctx or default_context(device_id)
        self.params = params
        self.slots = [first] + [Context(first.device_id) for _ in range(depth - 1)]
        for c in self.slots:
            _prepare_windows(c, params.stft_size, params.stft_shift, window)
            # (with utterances in flight beside each other a call stays on its own stream)
            c.set_utterances_in_flight(depth)
        self._bufs = [dict() for _ in self.slots]
        self._pending = deque()
        self._next = 0
        self.dropped = []       # tags close() drained without handing out their result
        # page-locked input sets for enqueue_staged (acquire_staging blocks until one is free;
        # a set is free again once its utterance was popped)
        self._staging = queue.Queue()
        self._all_staging = [HostStaging(first) for _ in range(staging_sets)]
        for st in self._all_staging:
            self._staging.put(st)
        self._out_host = [None for _ in self.slots]
        self._out_retired = []

    def __len__(self):
        return len(self._pending)

    def full(self):
        return len(self._pending) == len(self.slots)

    def _buffer(self, slot, name, nbytes):
        buf = self._bufs[slot].get(name)
        if buf is None or buf.nbytes < nbytes:
            buf = self._bufs[slot][name] = self.slots[slot].empty(max(int(nbytes * 1.25), 16))
        return buf

    def enqueue(self, tag, obs, activity, target_index, start_context, end_context):
        assert not self.full(), 'pop() the oldest utterance first'
        slot = self._next
        self._next = (self._next + 1) % len(self.slots)
        ctx, p = self.slots[slot], self.params
        # int16 = PCM straight from the WAV files: converted on the device
        pcm = np.asarray(obs).dtype == np.int16
        obs = np.ascontiguousarray(obs, dtype=np.int16 if pcm else np.float64)
        act = np.ascontiguousarray((np.asarray(activity) != 0).astype(np.uint8))
        D, N = obs.shape
        K, N_act = act.shape
        T = stft_frames(N, p.stft_size, p.stft_shift, p.stft_fading)
        n_out = int(ctx.lib.gss_istft_num_samples(T, p.stft_size, p.stft_shift, p.stft_fading))
        obs_d = self._buffer(slot, 'obs', obs.nbytes)
        act_d = self._buffer(slot, 'act', act.nbytes)
        out_d = self._buffer(slot, 'out', 8 * max(n_out, 1))
        ctx.upload(obs_d, obs)
        ctx.upload(act_d, act)
        entry = ctx.lib.gss_enhance_observation_pcm16 if pcm else ctx.lib.gss_enhance_observation
        ctx._check(entry(
            ctx.handle, ctypes.byref(p), c_void_p(obs_d.ptr), D, N, c_void_p(act_d.ptr), K,
            N_act, int(target_index), int(start_context), int(end_context),
            c_void_p(out_d.ptr), None), 'gss_enhance_observation')
        self._pending.append((tag, slot, n_out))

    # ---- page-locked staging (the session driver's path) ----------------------------
    def acquire_staging(self, timeout=None):
        """A free HostStaging set; blocks while all are in use (loader threads call this)."""
        return self._staging.get(timeout=timeout)

    def release_staging(self, staging):
        self._staging.put(staging)

    def enqueue_staged(self, tag, staging, target_index, start_context, end_context, keep=None):
        """Like enqueue() for inputs sitting in a HostStaging set: two asynchronous DMAs, the
        kernels behind them, and an asynchronous D2H of the samples ``keep = (a, b)`` of the
        result (default: all) into page-locked memory -- the host thread does not wait for any
        of it.  The set goes back to the free list when the utterance is popped."""
        assert not self.full(), 'pop() the oldest utterance first'
        slot = self._next
        self._next = (self._next + 1) % len(self.slots)
        ctx, p = self.slots[slot], self.params
        obs, act = staging.obs, staging.act
        D, N = obs.shape
        K, N_act = act.shape
        T = stft_frames(N, p.stft_size, p.stft_shift, p.stft_fading)
        n_out = int(ctx.lib.gss_istft_num_samples(T, p.stft_size, p.stft_shift, p.stft_fading))
        a, b = (0, n_out) if keep is None else (min(max(int(keep[0]), 0), n_out),
                                                 min(max(int(keep[1]), 0), n_out))
        b = max(a, b)
        obs_d = self._buffer(slot, 'obs', obs.nbytes)
        act_d = self._buffer(slot, 'act', act.nbytes)
        out_d = self._buffer(slot, 'out', 8 * max(n_out, 1))
        out_h = self._out_host[slot]
        if out_h is None or out_h.nbytes < 8 * (b - a):
            have = 0
            if out_h is not None:            # parked until close(), see HostStaging.shape
                self._out_retired.append(out_h)
                have = out_h.nbytes
            out_h = self._out_host[slot] = ctx.pinned(
                max(int(8 * (b - a) * 1.25) + 4096, 2 * have))
        out_view = out_h.view((b - a,), np.float64)
        ctx.upload_async(obs_d, obs)
        ctx.upload_async(act_d, act)
        ctx._check(ctx.lib.gss_enhance_observation_pcm16(
            ctx.handle, ctypes.byref(p), c_void_p(obs_d.ptr), D, N, c_void_p(act_d.ptr), K,
            N_act, int(target_index), int(start_context), int(end_context),
            c_void_p(out_d.ptr), None), 'gss_enhance_observation')
        if b > a:
            ctx.download_async(out_view, out_d, offset=8 * a)
        self._pending.append((tag, slot, n_out, staging, out_view))

    def pop(self):
        entry = self._pending.popleft()
        tag, slot, n_out = entry[:3]
        if len(entry) == 5:
            staging, out_view = entry[3:]
            try:
                self.slots[slot].synchronize()
                x_hat = out_view.copy()      # the slot's pinned block is reused by the next one
            finally:
                self.release_staging(staging)
        else:
            x_hat = self.slots[slot].to_host(self._bufs[slot]['out'], (n_out,), np.float64)
        if self.params.bf in (_BF_CODES['mvdrSouden_ban'], _BF_CODES['gev_ban']):
            # what the reference raises for this utterance (raised explicitly: an `assert`
            # statement disappears under python -O and NaN audio would be written silently);
            # `tag` may be a whole example dict: name it by its id.
            name = tag.get('example_id', '?') if isinstance(tag, dict) else tag
            _raise_for_ref_channel(self.slots[slot].last_ref_channel(), name)
        return tag, x_hat

    def close(self):
        """Drain what is still in flight WITHOUT the status check (close() runs in `finally`
        blocks: a second exception here would mask the first one and leak the extra
        contexts), then release buffers and contexts.  Consequence for a caller that enqueues
        and closes without pop(): the AssertionError / LinAlgError the reference would raise
        for those utterances (non-finite SNR, indefinite noise PSD) is NOT raised -- their
        results are discarded unseen.  pop() every utterance whose outcome matters
        (`Enhancer._enhance_and_write` does); `self.dropped` lists the tags close() drained."""
        try:
            while self._pending:
                tag, slot = self._pending.popleft()[:2]
                self.dropped.append(tag)
                try:
                    self.slots[slot].synchronize()
                except Exception:
                    pass
        finally:
            self._bufs = [dict() for _ in self.slots]
            for block in (self._all_staging + self._out_retired
                          + [h for h in self._out_host if h is not None]):
                try:
                    block.free()
                except Exception:
                    pass
            self._all_staging, self._out_host = [], [None for _ in self.slots]
            self._out_retired = []
            for c in self.slots[1:]:
                try:
                    c.close()
                except Exception:
                    pass
            self.slots = self.slots[:1]
            try:        # the first slot is the caller's context: nothing said again
                self.slots[0].set_utterances_in_flight(0)
            except Exception:
                pass


def enhance_observation(obs, activity, target_index, start_context_samples,
                        end_context_samples, *, params=None, window=None, debug=False,
                        ctx=None, **param_kwargs):
    """Fused per-utterance pipeline (core.py:514-571), intermediates resident in HBM.

    obs (D,N) float64, activity (K,N) bool in dict order.  Returns x_hat, or
    (x_hat, details) with ``debug=True`` where details holds the reference's
    debug locals in the reference's layouts."""
    ctx = ctx or default_context()
    if params is None:
        params = make_params(**param_kwargs)
    _prepare_windows(ctx, params.stft_size, params.stft_shift, window)
    utt = ResidentUtterance(ctx, obs, activity, params)
    D, K, T = utt.D, utt.K, utt.T
    F = params.stft_size // 2 + 1
    taps = None
    bufs = {}
    if debug:
        bufs = {
            'Obs_ftd': ctx.empty(16 * F * T * D), 'act_frames': ctx.empty(max(K * T, 16)),
            'gamma': ctx.empty(8 * F * K * T), 'target_mask': ctx.empty(8 * F * T),
            'distortion_mask': ctx.empty(8 * F * T), 'Xhat': ctx.empty(16 * F * T),
            'ref_channel': ctx.empty(16),
        }
        ctx._check(ctx.lib.gss_memset(ctx.handle, c_void_p(bufs['ref_channel'].ptr), 0xFF, 16),
                   'gss_memset')
        taps = GssDebugTaps(**{k: v.ptr for k, v in bufs.items()})
    utt.enqueue(target_index, start_context_samples, end_context_samples, taps)
    x_hat = utt.result()
    if not debug:
        return x_hat
    details = {
        'Obs': _ftd_to_host_dtf(ctx, bufs['Obs_ftd'], D, T, F),
        'acitivity_freq': ctx.to_host(bufs['act_frames'], (K, T), np.uint8).astype(bool),
        'posterior': ctx.to_host(bufs['gamma'], (F, K, T), np.float64).transpose(1, 2, 0),
        'target_mask': ctx.to_host(bufs['target_mask'], (F, T), np.float64).T,
        'distortion_mask': ctx.to_host(bufs['distortion_mask'], (F, T), np.float64).T,
        'X_hat': ctx.to_host(bufs['Xhat'], (T, F), np.complex128),
        'ref_channel': int(ctx.to_host(bufs['ref_channel'], (1,), np.int32)[0]),
    }
    return x_hat, details
