"""Slices of PCM16 WAV files read straight into caller-owned int16 rows.

The reference loads every channel file of every array with ``soundfile`` (libsndfile opens the
file, seeks, decodes to float64), stacks the channels with ``np.array`` and reshapes
(/root/reference/pb_chime5/io/audioread.py:34-226, core.py:427-470): for a 24-channel example
with 2 x 15 s of context that is 24 opens + 24 float conversions + two copies of 100 MB per
utterance.  The session driver here needs the PCM samples untouched (the STFT kernel does the
``/ 32768``), in one (D, N) int16 block of page-locked memory that a single DMA takes to the
GPU.  So: the RIFF header of a file is parsed once and its descriptor kept open; a slice is one
``preadv`` of ``2 * (stop - start)`` bytes into the row it belongs to -- no intermediate buffer,
no GIL while the kernel copies.  CHiME-5 / CHiME-6 ship one mono file per microphone
(``S02_U01.CH1.wav``), which is the fast path; a file with several channels is de-interleaved
through a scratch read.

Same conventions as ``load_audio(path, start, stop)``: ``stop`` is clipped to the file,
``start`` beyond the end gives an empty slice, only 16-bit PCM is supported
(NotImplementedError otherwise).
"""
import os
import struct
import threading
from collections import OrderedDict

import numpy as np


class WavInfo:
    """`users` / `evicted` belong to the reader's lock: a descriptor is closed when it has left
    the LRU table AND the last read on it has returned -- never under a thread inside `preadv`
    (the number of a closed descriptor is handed to the next `open`; a read on it would return
    another file's samples without any error)."""
    __slots__ = ('fd', 'data_offset', 'frames', 'channels', 'sample_rate', 'users', 'evicted')

    def __init__(self, fd, data_offset, frames, channels, sample_rate):
        self.fd, self.data_offset, self.frames = fd, data_offset, frames
        self.channels, self.sample_rate = channels, sample_rate
        self.users, self.evicted = 0, False


def parse_wav_header(fd, path='<fd>'):
    """Walk the RIFF chunks of an open file: (data offset, frames, channels, sample rate)."""
    head = os.pread(fd, 12, 0)
    if len(head) < 12 or head[:4] != b'RIFF' or head[8:12] != b'WAVE':
        raise ValueError(f'{path}: not a RIFF/WAVE file')
    pos = 12
    fmt = None
    size = os.fstat(fd).st_size
    while pos + 8 <= size:
        cid, clen = struct.unpack('<4sI', os.pread(fd, 8, pos))
        body = pos + 8
        if cid == b'fmt ':
            raw = os.pread(fd, min(clen, 40), body)
            if len(raw) < 16:
                raise ValueError(f'{path}: fmt chunk of {len(raw)} bytes')
            tag, channels, rate, _, block_align, bits = struct.unpack('<HHIIHH', raw[:16])
            if tag == 0xFFFE and len(raw) >= 26:          # WAVE_FORMAT_EXTENSIBLE: sub-format
                tag = struct.unpack('<H', raw[24:26])[0]
            if tag != 1 or bits != 16:
                raise NotImplementedError(
                    f'{path}: only 16-bit PCM is supported, got format {tag}, {bits} bit')
            if channels < 1 or block_align != 2 * channels:
                raise ValueError(f'{path}: {channels} channels of 16 bit with block alignment '
                                 f'{block_align}')
            fmt = (channels, rate)
        elif cid == b'data':
            if fmt is None:
                raise ValueError(f'{path}: data chunk before fmt chunk')
            channels, rate = fmt
            clen = min(clen, size - body)                  # truncated / streamed files
            return body, clen // (2 * channels), channels, rate
        pos = body + clen + (clen & 1)
    raise ValueError(f'{path}: no data chunk')


class WavSliceReader:
    """Keeps up to `max_open` files open (LRU); thread-safe."""

    def __init__(self, max_open=256):
        self._open = OrderedDict()
        self._lock = threading.Lock()
        self.max_open = max_open

    def info(self, path, pin=False):
        """The header record of `path`.  ``pin=True`` (every caller that goes on to use
        ``info.fd``) holds the descriptor open until the matching `_unpin`."""
        key = os.fspath(path)
        with self._lock:
            info = self._open.get(key)
            if info is not None:
                self._open.move_to_end(key)
                info.users += pin
                return info
        fd = os.open(key, os.O_RDONLY)
        try:
            info = WavInfo(fd, *parse_wav_header(fd, key))
        except Exception:
            os.close(fd)
            raise
        with self._lock:
            other = self._open.get(key)
            if other is not None:            # another thread was faster
                os.close(fd)
                other.users += pin
                return other
            self._open[key] = info
            info.users += pin
            while len(self._open) > self.max_open:
                _, old = self._open.popitem(last=False)
                old.evicted = True
                if old.users == 0:
                    os.close(old.fd)
        return info

    def _unpin(self, info):
        with self._lock:
            info.users -= 1
            if info.evicted and info.users == 0:
                os.close(info.fd)

    def slice_length(self, path, start=None, stop=None):
        """Number of sample frames ``load_audio(path, start, stop)`` returns."""
        total = self.info(path).frames
        start = 0 if start is None else min(int(start), total)
        stop = total if stop is None else min(int(stop), total)
        return max(stop - start, 0)

    def read_into(self, path, start, out):
        """Fill ``out`` -- int16, (n,) for a mono file or (channels, n) -- with the ``n`` sample
        frames from ``start`` on.  The caller has clipped ``n`` with `slice_length`."""
        n = out.shape[-1]
        if n == 0:
            return
        info = self.info(path, pin=True)
        try:
            start = 0 if start is None else int(start)
            assert out.dtype == np.int16 and start + n <= info.frames, \
                (path, start, n, info.frames)
            if info.channels == 1:
                row = out.reshape(-1)
                assert row.flags.c_contiguous and row.shape[0] == n
                self._pread_all(info.fd, memoryview(row).cast('B'),
                                info.data_offset + 2 * start)
                return
            assert out.shape == (info.channels, n), (out.shape, info.channels, n)
            scratch = np.empty((n, info.channels), dtype='<i2')
            self._pread_all(info.fd, memoryview(scratch).cast('B'),
                            info.data_offset + 2 * info.channels * start)
            out[...] = scratch.T
        finally:
            self._unpin(info)

    @staticmethod
    def _pread_all(fd, view, offset):
        done = 0
        while done < len(view):
            got = os.preadv(fd, [view[done:]], offset + done)
            if got <= 0:
                raise EOFError(f'short read at byte {offset + done}')
            done += got

    def close(self):
        with self._lock:
            for info in self._open.values():
                info.evicted = True
                if info.users == 0:
                    os.close(info.fd)
            self._open.clear()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
