"""RTTM-driven example source for CHiME-6 style directories: the front door of the
reference's track-2 path (/root/reference/pb_chime5/core_chime6_rttm.py:288-357,
database/chime5/rttm.py:285-632, utils/intervall_array.py:45-101).

* ``from_rttm`` parses ``SPEAKER <file> <chn> <begin> <dur> <NA> <NA> <name> <NA>``
  lines into ``{file_id: {speaker: ArrayIntervall}}`` with decimal-exact
  seconds -> samples conversion (the reference asserts the products are integers).
* ``RTTMDatabase`` enumerates one example per (session, speaker, interval) with the
  Kaldi-compatible example id ``S02_U06.-P05-000123456_000234567``, adds context
  (``backup_orig_start_end`` + ``AddContext`` for scalar start / end,
  database.py:706-710, 965-1011) and loads the audio cut to the shortest file.

Everything here is integer / string logic and host IO; the arithmetic of an example
runs in ``core_chime6_rttm.Enhancer.enhance_observation`` on the GPU.
"""
import collections
import decimal
from pathlib import Path

import numpy as np

from pb_chime5_amd.io import load_audio
from pb_chime5_amd.utils.intervall_array import ArrayIntervall


class _Constant:
    """paderbox ``array_intervall.zeros()`` / ``ones()`` without a shape: every slice
    is all False / all True (the reference's garbage class, core_chime6_rttm.py:58-65)."""

    def __init__(self, value):
        self.value = bool(value)
        self.shape = None

    def __getitem__(self, item):
        assert isinstance(item, slice) and item.step is None, item
        assert item.start is not None and item.stop is not None, item
        return np.full(max(item.stop - item.start, 0), self.value, dtype=bool)

    def __repr__(self):
        return f'{"ones" if self.value else "zeros"}(shape=None)'


def zeros():
    return _Constant(False)


def ones():
    return _Constant(True)


def from_rttm(rttm_file, shape=None, sample_rate=16000):
    """One or several RTTM files -> {file_id: {speaker: ArrayIntervall}}."""
    if isinstance(rttm_file, (str, Path)):
        rttm_file = [rttm_file]
    data = collections.defaultdict(lambda: collections.defaultdict(lambda: ArrayIntervall(shape)))
    for path in rttm_file:
        for line in Path(path).read_text().splitlines():
            parts = line.split()
            if not parts:
                continue
            assert parts[0] == 'SPEAKER', line
            file_id, name = parts[1], parts[7]
            begin = decimal.Decimal(parts[3])
            end = (begin + decimal.Decimal(parts[4])) * sample_rate
            begin = begin * sample_rate
            assert begin == int(begin), (line, begin)
            assert end == int(end), (line, end)
            data[file_id][name][int(begin):int(end)] = 1
    return {k: dict(v) for k, v in data.items()}


def strip_file_id(rttm):
    """The Kaldi recipes append ``_U06`` / ``.ENH`` to the session id in RTTM files."""
    out = {k.replace('_U06', '').replace('.ENH', ''): v for k, v in rttm.items()}
    assert len(out) == len(rttm), (tuple(out), tuple(rttm))
    return out


def get_chime6_files(chime6_dir, worn=False, flat=False):
    """``audio/<set>/S02_U01.CH1.wav`` ... -> {session: {array: [4 files]}} (or a flat
    list per session, or {session: {speaker: file}} for the worn microphones)."""
    chime6_dir = Path(chime6_dir)
    if worn:
        assert flat is False, flat
        files = sorted(chime6_dir.glob('audio/*/*_P*.wav'))
        out = collections.defaultdict(dict)
        for p in files:
            session, rest = p.name.split('_')[:2]
            out[session][rest.split('.')[0]] = str(p)
    else:
        files = sorted(chime6_dir.glob('audio/*/*_U*.wav'))
        if flat:
            out = collections.defaultdict(list)
            for p in files:
                out[p.name.split('_')[0]].append(str(p))
        else:
            out = collections.defaultdict(lambda: collections.defaultdict(list))
            for p in files:
                session, rest = p.name.split('_')[:2]
                out[session][rest.split('.')[0]].append(str(p))
            out = {k: dict(v) for k, v in out.items()}
    assert len(files) > 0, (files, chime6_dir)
    return dict(out)


def select_channels(chime6_dir, multiarray):
    """core_chime6_rttm.get_database's channel selection (:313-347)."""
    if multiarray is True:
        return get_chime6_files(chime6_dir, worn=False, flat=True)
    files = get_chime6_files(chime6_dir, worn=False, flat=False)
    if multiarray == 'outer_array_mics':
        return {s: [f for arr in arrays.values() for f in (arr[0], arr[-1])]
                for s, arrays in files.items()}
    if multiarray == 'first_array_mics':
        return {s: [arr[0] for arr in arrays.values()] for s, arrays in files.items()}
    raise ValueError(multiarray)


def recursive_load_audio(paths, start=0, stop=None, min_num_samples=1, dtype=np.float64):
    """Load every channel file, drop the ones that end before the segment ("last 15
    minutes of U05 missing"), cut the rest to the shortest (rttm.py:550-632)."""
    data = [load_audio(p, start=start, stop=stop, dtype=dtype) for p in paths]
    kept = [d for d in data if d.shape[-1] >= min_num_samples]
    assert len(kept) >= len(data) - 8, (len(kept), len(data))
    num_samples = min(d.shape[-1] for d in kept)
    assert num_samples >= min_num_samples, (num_samples, min_num_samples)
    return np.array([d[..., :num_samples] for d in kept])


def split_context(samples):
    if isinstance(samples, (tuple, list)):
        if len(samples) == 1:
            samples = (samples[0], samples[0])
        start, end = samples
    else:
        start = end = samples
    assert start >= 0 and end >= 0, f'Negative context value ({samples}) is not supported'
    return int(start), int(end)


class RTTMDatabase:
    def __init__(self, rttm_path, audio_paths, alias=None):
        self._rttm_path = rttm_path
        self._audio_paths = audio_paths
        self._alias = alias or {}
        self._rttm = strip_file_id(from_rttm(rttm_path))

    @staticmethod
    def example_id(file_id, speaker_id, start, end):
        max_digits = len(str(16000 * 60 * 60 * 10))      # 10 h of samples
        return (f'{file_id}_U06.-{speaker_id}-{str(start).zfill(max_digits)}'
                f'_{str(end).zfill(max_digits)}')

    @property
    def dataset_names(self):
        return tuple(self._rttm) + tuple(self._alias)

    def get_examples(self, session):
        sessions = []
        for s in ((session,) if isinstance(session, str) else tuple(session)):
            sessions.extend(self._alias.get(s, (s,)))
        out = []
        for session_id in sessions:
            # One example per RTTM line, in file order per speaker: the reference walks
            # ``speaker.intervals`` (rttm.py:466), the segments as they were assigned -- NOT
            # merged or sorted, so two touching segments of a speaker stay two utterances --
            # into a dict keyed by the example id (a repeated line is one example).
            examples = {}
            for speaker_id, speaker in self._rttm[session_id].items():
                for start, end in speaker.intervals:
                    example_id = self.example_id(session_id, speaker_id, start, end)
                    examples[example_id] = {
                        'example_id': example_id,
                        'start': start, 'end': end, 'num_samples': end - start,
                        'session_id': session_id, 'speaker_id': speaker_id,
                        'audio_path': self._audio_paths[session_id], 'dataset': session_id,
                    }
            out.extend(examples.values())
        return out

    def get_dataset_for_session(self, session, *, audio_read=False, adjust_times=False,
                                context_samples=0, equal_start_context=False):
        assert adjust_times is False, 'not necessary for CHiME-6 (synchronised arrays)'
        examples = self.get_examples(session)
        if context_samples != 0:
            start_context, end_context = split_context(context_samples)
            for ex in examples:
                ex['start_orig'], ex['end_orig'] = ex['start'], ex['end']
                ex['num_samples_orig'] = ex['num_samples']
                ex['start'] = max(ex['start'] - start_context, 0)
                ex['end'] = ex['end'] + end_context
                ex['num_samples'] = ex['end'] - ex['start']
        if audio_read is True:
            return LazyAudio(examples)
        assert audio_read is False, audio_read
        return examples


class LazyAudio:
    """Sequence of examples whose ``audio_data`` is read when an item is accessed."""

    def __init__(self, examples, dtype=np.float64):
        self.examples = examples
        self.dtype = dtype       # np.int16: PCM as stored (converted on the device)

    def __len__(self):
        return len(self.examples)

    def __getitem__(self, item):
        if isinstance(item, slice):
            return LazyAudio(self.examples[item], self.dtype)
        ex = dict(self.examples[item])
        min_num_samples = ex.get('end_orig', ex['end']) - ex['start']
        ex['audio_data'] = recursive_load_audio(
            ex['audio_path'], start=ex['start'], stop=ex['end'],
            min_num_samples=min_num_samples, dtype=self.dtype)
        return ex

    def __iter__(self):
        return (self[i] for i in range(len(self)))
