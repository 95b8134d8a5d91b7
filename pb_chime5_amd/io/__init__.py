"""Minimal WAV in/out with the conventions of the reference's io layer
(/root/reference/pb_chime5/io/audioread.py:34-226, io/audiowrite.py:16-207):

* ``load_audio(path, start=None, stop=None)`` returns float64 in [-1, 1)
  (PCM16 / 32768), shape (samples,) for mono and (channels, samples) otherwise;
  lists / tuples / dicts of paths are loaded recursively and lists are stacked
  into one array (the reference's ``recursive_load_decorator``).  ``dtype=np.int16``
  (an addition) hands out the PCM samples untouched: the session driver uploads those
  and lets the STFT kernel do the ``/ 32768`` -- a quarter of the bytes, no float64 copy.
* ``dump_audio(obj, path)`` peak-normalises to (2**15 - 1) / 2**15 and writes
  16-bit PCM at 16 kHz, like ``dump_audio(..., normalize=True, dtype=np.int16)``.
  The float -> PCM16 step happens inside libsndfile in the reference; its result is
  pinned by the doctest at io/audiowrite.py:49-52 ([1, 2, -4, 4] normalised reads back
  as [0.24996948, 0.49996948, -0.99996948, 0.99996948], i.e. 8191.75 -> 8191 and
  16383.5 -> 16383): scale to 32 bit, round, keep the upper 16 bits (floor), saturate
  [UPSTREAM-RECALL libsndfile's clipping double -> short conversion].

The reference uses ``soundfile`` (absent in this image); only RIFF/WAVE PCM16 --
what CHiME-5/6 ships -- is handled here: a small RIFF chunk walk (io/wav_slices.py, shared with
the session driver's slice reader) for reading, the standard-library ``wave`` for writing.
"""
import wave
from pathlib import Path

import numpy as np


def _load_one(path, start=None, stop=None, dtype=np.float64):
    import os
    from pb_chime5_amd.io.wav_slices import WavSliceReader, parse_wav_header
    fd = os.open(os.fspath(path), os.O_RDONLY)
    try:
        offset, total, channels, _ = parse_wav_header(fd, path)   # NotImplementedError: not PCM16
        start = 0 if start is None else min(int(start), total)
        stop = total if stop is None else min(int(stop), total)
        count = max(stop - start, 0)
        raw = bytearray(2 * channels * count)
        WavSliceReader._pread_all(fd, memoryview(raw), offset + 2 * channels * start)
    finally:
        os.close(fd)
    data = np.frombuffer(raw, dtype='<i2')
    if np.dtype(dtype) != np.int16:
        data = data.astype(dtype) / 2 ** 15
    if channels == 1:
        return data
    return data.reshape(-1, channels).T


def load_audio(path, start=None, stop=None, dtype=np.float64):
    if isinstance(path, dict):
        return {k: load_audio(v, start=start, stop=stop, dtype=dtype) for k, v in path.items()}
    if isinstance(path, (list, tuple)):
        return np.array([load_audio(p, start=start, stop=stop, dtype=dtype) for p in path])
    return _load_one(path, start=start, stop=stop, dtype=dtype)


def dump_audio(obj, path, *, sample_rate=16000, normalize=True):
    obj = np.asarray(obj)
    if normalize:
        if obj.dtype.kind not in 'fi':
            raise TypeError(f'Only float and int is supported with normalize, got {obj.dtype}')
        correction = (2 ** 15 - 1) / (2 ** 15)
        obj = obj * (correction / np.amax(np.abs(obj)))
    if obj.dtype.kind == 'f':
        wide = np.rint(np.clip(obj.astype(np.float64), -1.0, 1.0) * 2.0 ** 31)
        pcm = np.clip(np.floor(wide / 2 ** 16), -2 ** 15, 2 ** 15 - 1).astype('<i2')
    else:
        pcm = obj.astype('<i2')
    channels = 1 if pcm.ndim == 1 else pcm.shape[0]
    Path(path).parent.mkdir(parents=True, exist_ok=True)
    with wave.open(str(path), 'wb') as w:
        w.setnchannels(channels)
        w.setsampwidth(2)
        w.setframerate(sample_rate)
        w.writeframes(np.ascontiguousarray(pcm.T).tobytes())
