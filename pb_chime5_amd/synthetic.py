"""Seeded synthetic CHiME-5-shaped utterances.

There is no corpus in the build or bench environment, so every config of
BASELINE.json is exercised on synthetic multi-channel recordings of the same
shape: coloured-noise "speech" sources with 4 Hz amplitude modulation, one random
decaying room impulse response per (source, microphone) so that WPE has
reverberation to remove, sensor noise, and per-speaker activity intervals plus an
always-active ``Noise`` class (the reference default ``activity_garbage_class``,
/root/reference/pb_chime5/core.py:587).

The same bytes are produced on every machine for a given seed, so CPU oracle and
GPU path see identical inputs.
"""
from dataclasses import dataclass, field

import numpy as np

SAMPLE_RATE = 16000


@dataclass
class Utterance:
    obs: np.ndarray            # (D, N) float64
    activity: dict             # speaker id -> bool (N,), insertion ordered
    speaker_id: str            # target speaker
    ex: dict                   # the ``ex`` keys enhance_observation consumes
    seconds: float = field(default=0.0)

    @property
    def activity_array(self):
        return np.array(list(self.activity.values()))

    @property
    def target_index(self):
        return tuple(self.activity.keys()).index(self.speaker_id)


def _source(rng, n):
    from scipy.signal import lfilter
    x = lfilter([1.0], [1.0, -0.9], rng.standard_normal(n))
    t = np.arange(n) / SAMPLE_RATE
    x = x * (0.6 + 0.4 * np.sin(2 * np.pi * 4.0 * t + rng.uniform(0, 2 * np.pi)))
    return x / np.sqrt(np.mean(x ** 2))


def _rir(rng, taps=2048):
    t = np.arange(taps) / SAMPLE_RATE
    h = rng.standard_normal(taps) * np.exp(-t / 0.05) * 0.1
    delay = int(rng.integers(0, 41))
    h[:delay] = 0.0
    h[delay] = 1.0
    return h


def _reverberate_fast(src, rirs, num_samples):
    """Sum over sources of src[k] * rirs[k, d] for all microphones with one batched FFT
    (the sum is taken in the frequency domain: K + K*D + D transforms instead of K*D
    full convolutions).  Same signal as the per-pair ``fftconvolve`` loop up to
    rounding (~1e-16 relative), ~20x faster; used where the bytes need not equal the
    slow path's (bench workloads)."""
    import scipy.fft as sfft
    K, D, taps = rirs.shape
    nfft = sfft.next_fast_len(num_samples + taps - 1, real=True)
    S = sfft.rfft(src, nfft, axis=-1, workers=-1)                    # (K, nf)
    out = np.zeros((D, nfft // 2 + 1), dtype=np.complex128)
    for k in range(K):
        out += S[k] * sfft.rfft(rirs[k], nfft, axis=-1, workers=-1)
    return sfft.irfft(out, nfft, axis=-1, workers=-1)[:, :num_samples]


def make_utterance(seed, num_channels, num_samples, intervals, target=0,
                   start_context=0, end_context=0, rir_taps=2048, noise=1e-3, fast=False,
                   diffuse_noise=0.0):
    """intervals: list of (start, stop) sample pairs, one per speaker; the
    ``Noise`` class is appended as all-True.  ``start_context``/``end_context``
    are the context samples on each side of the core segment.

    ``diffuse_noise`` > 0 adds a spatially diffuse background (an independent coloured
    noise of that RMS, relative to one speech source, at every microphone), which keeps
    the distortion PSD matrix well conditioned in every frequency bin; the SURVEY 8d
    generator (``diffuse_noise=0``, white sensor noise only) leaves it nearly singular
    wherever a few frames carry the whole distortion mask."""
    from scipy.signal import fftconvolve
    rng = np.random.default_rng(seed)
    obs = np.zeros((num_channels, num_samples))
    activity = {}
    srcs, rirs = [], []
    for k, (a, b) in enumerate(intervals):
        act = np.zeros(num_samples, dtype=bool)
        act[a:b] = True
        activity[f'P{k + 1:02d}'] = act
        src = _source(rng, num_samples) * act
        if fast:
            srcs.append(src)
            rirs.append(np.stack([_rir(rng, rir_taps) for _ in range(num_channels)]))
            continue
        for d in range(num_channels):
            obs[d] += fftconvolve(src, _rir(rng, rir_taps))[:num_samples]
    if fast and srcs:
        obs += _reverberate_fast(np.stack(srcs), np.stack(rirs), num_samples)
    activity['Noise'] = np.ones(num_samples, dtype=bool)
    obs += rng.standard_normal(obs.shape) * noise
    if diffuse_noise > 0:
        for d in range(num_channels):
            obs[d] += _source(rng, num_samples) * diffuse_noise
    obs *= 0.1
    speaker_id = f'P{target + 1:02d}'
    ex = {
        'start': {'original': 0},
        'start_orig': {'original': start_context},
        'end_orig': {'original': num_samples - end_context},
        'end': {'original': num_samples},
        'speaker_id': speaker_id,
    }
    return Utterance(obs, activity, speaker_id, ex, num_samples / SAMPLE_RATE)


def config1(seed=1, context=0):
    """BASELINE.json configs[0]: 4 mics, 5 s, 2 speakers (+Noise), WPE off,
    5 EM iterations."""
    sr = SAMPLE_RATE
    return make_utterance(
        seed, 4, 80000,
        [(int(0.5 * sr), int(4.0 * sr)), (int(2.0 * sr), int(4.5 * sr))],
        target=0, start_context=context, end_context=context)


def config2(seed=2, num_channels=24, seconds=15.0, num_speakers=4):
    """BASELINE.json configs[1]: 24 mics (6 arrays x 4), 15 s, 4 speakers
    (+Noise), 5 s core with 5 s context each side; speakers 30-60 % active."""
    sr = SAMPLE_RATE
    n = int(seconds * sr)
    rng = np.random.default_rng(seed + 7919)
    ctx = n // 3
    intervals = [(ctx, n - ctx)]        # target is active over the core
    for _ in range(num_speakers - 1):
        length = int(rng.uniform(0.3, 0.6) * n)
        a = int(rng.integers(0, n - length))
        intervals.append((a, a + length))
    # the target speaks 30-60 % too: extend its interval around the core
    extra = int(rng.uniform(0.0, 0.25) * n)
    a = max(ctx - extra // 2, 0)
    intervals[0] = (a, min(n - ctx + extra // 2, n))
    return make_utterance(seed, num_channels, n, intervals, target=0,
                          start_context=ctx, end_context=ctx)


def config3_core_samples(index):
    """Core (utterance proper, without context) length in samples of dev-shaped item
    ``index`` of BASELINE.json configs[2]: ~ LogNormal(ln 2.5 s, 0.7) clipped to
    [0.5 s, 15 s], first draw of the item's generator (seed 1000 + index + 104729)."""
    rng = np.random.default_rng(1000 + index + 104729)
    return int(np.clip(rng.lognormal(np.log(2.5), 0.7), 0.5, 15.0) * SAMPLE_RATE)


def config3_item(index, num_channels=24, context=240000):
    """BASELINE.json configs[2]: dev-shaped utterance number ``index`` (seed
    1000 + index); core length ~ LogNormal(ln 2.5 s, 0.7) clipped to
    [0.5 s, 15 s], reference-default context of 240000 samples each side."""
    sr = SAMPLE_RATE
    seed = 1000 + index
    rng = np.random.default_rng(seed + 104729)
    core = int(np.clip(rng.lognormal(np.log(2.5), 0.7), 0.5, 15.0) * sr)
    assert core == config3_core_samples(index)
    n = core + 2 * context
    intervals = [(context, context + core)]
    for _ in range(3):
        length = int(rng.uniform(0.3, 0.6) * n)
        a = int(rng.integers(0, n - length))
        intervals.append((a, a + length))
    # CHiME-5 is a dinner party: somebody else talks over the target.  One interferer is
    # moved so that it covers part of the core (otherwise the distortion mask is empty
    # inside the only frames the beamformer looks at and its PSD matrix has rank 1), and
    # the sensor noise sits 30 dB below the speech instead of 60 dB.
    length = intervals[1][1] - intervals[1][0]
    a = int(np.clip(context + core // 2 - length // 2 + rng.integers(-sr, sr + 1), 0, n - length))
    intervals[1] = (a, a + length)
    return make_utterance(seed, num_channels, n, intervals, target=0,
                          start_context=context, end_context=context,
                          rir_taps=1024, noise=3e-2)


def config5(seed=5, num_channels=12, seconds=120.0):
    """BASELINE.json configs[4]: 120 s segment, 'outer_array_mics' -> 12 ch."""
    sr = SAMPLE_RATE
    n = int(seconds * sr)
    rng = np.random.default_rng(seed + 15485863)
    ctx = int(50 * sr)
    intervals = [(ctx, n - ctx)]
    for _ in range(3):
        length = int(rng.uniform(0.3, 0.6) * n)
        a = int(rng.integers(0, n - length))
        intervals.append((a, a + length))
    return make_utterance(seed, num_channels, n, intervals, target=0,
                          start_context=ctx, end_context=ctx)


def tiny(seed=0, num_channels=4, num_samples=12000, num_speakers=2,
         context=2048, noise=1e-3):
    """Small case for smoke tests and CPU-sized parity checks."""
    n = num_samples
    rng = np.random.default_rng(seed + 31)
    intervals = [(context, n - context)]
    for _ in range(num_speakers - 1):
        length = int(rng.uniform(0.3, 0.6) * n)
        a = int(rng.integers(0, n - length))
        intervals.append((a, a + length))
    return make_utterance(seed, num_channels, n, intervals, target=0,
                          start_context=context, end_context=context,
                          rir_taps=512, noise=noise)
