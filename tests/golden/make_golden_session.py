"""Golden fixture for the CHiME-5 JSON front door.

Runs ONLY in the build container (needs /root/reference).  It writes the synthetic
CHiME-5-layout corpus of ``pb_chime5_amd.synthetic_corpus`` to a scratch directory and
drives the reference's REAL session code on it:

* ``Chime5(json).get_iterator_for_session(...)`` (database.py:83-131: redacted filter,
  backup, adjust_start_end, AddContext with equal start context),
* ``get_activity(..., perspective='array', use_ArrayIntervall=True)`` (activity.py:8-222),
* the CHiME-6 twins of the above (core_chime6.py, activity.py:225-403) on the same
  corpus written with one synchronised clock,
* ``Enhancer.enhance_example`` (core.py:396-512) with the reference's own ``load_audio``
  (io/audioread.py:34-226) reading the WAV files through a minimal ``soundfile``
  stand-in built on the standard ``wave`` module,

with the numeric third-party calls delegated to the CPU oracle as in make_golden.py.
``lazy_dataset`` (absent here) is replaced by a small list-backed stand-in defined in
this file.  The reference's static table of recording lengths is pointed at the
synthetic recordings' length (it only bounds the activity tracks).

Output: chime5_session.json / chime6_session.json (bookkeeping, bit exact) and
chime5_session.npz / chime6_session.npz (enhanced signals), and chime5_dev_sessions.{json,npz}:
the same for ``session_id=dev`` = S02 (6 arrays) AND S09 (5 arrays, mapping.py:67) behind one
database -- the iterator over both sessions, the activity of each, examples enhanced with
``multiarray=True`` (24 and 20 channels).  Usage:  python tests/golden/make_golden_session.py
"""
import copy
import hashlib
import json
import sys
import tempfile
import types
import wave
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import make_golden as mg  # noqa: E402

CORPUS = dict(session_id='S02', seconds=9.0, seed=11, utts_per_speaker=2, num_redacted=1)
ENHANCER = dict(context_samples=16000, multiarray='outer_array_mics', wpe=True, wpe_tabs=4,
                wpe_iterations=2, bss_iterations=5)
EXAMPLES = (0, 5)
# dense overlap and diffuse noise: with all 24 microphones the noise PSD matrix needs more than 24
# frames' worth of distortion-mask mass inside the core of an utterance, or the reference's own
# beamformer output is rounding noise (cond(Phi_N) 1e16 on a sparser corpus; here <= 6e5)
DEV_CORPUS = dict(session_id=['S02', 'S09'], seconds=6.0, seed=23, utts_per_speaker=4, num_redacted=1,
                  noise=0.1)
DEV_ENHANCER = dict(context_samples=12000, multiarray=True, wpe=True, wpe_tabs=2,
                    wpe_iterations=2, bss_iterations=4)
# one of S02 (24 channels), one of S09 (20 channels), both well posed in the reference's own
# float64 arithmetic.  (Not all 32 are: in examples 0, 4, 9, 10, 21, 22, 26, 29 the WPE normal
# equations of the lowest bins are conditioned so badly that two float64 solves differ by
# 1e-6 ... 5e-5 after WPE already, and in example 18 one bin's distortion mask sums to 8 frames
# on 20 channels -- cond(Phi_N) = 2e18, the reference's output there is 1e10 and its reference
# channel a coin toss.  Measured with tools/fuzz_case.py-style stage comparisons in round 6.)
DEV_EXAMPLES = (2, 28)


# ---------------------------------------------------------------- stand-ins
class _Dataset:
    def __init__(self, examples, maps=()):
        self.examples = list(examples)
        self.maps = tuple(maps)

    def _get(self, ex):
        ex = copy.deepcopy(ex)
        for m in self.maps:
            ex = m(ex)
        return ex

    def __len__(self):
        return len(self.examples)

    def __iter__(self):
        return (self._get(ex) for ex in self.examples)

    def __getitem__(self, item):
        if isinstance(item, slice):
            return _Dataset(self.examples[item], self.maps)
        return self._get(self.examples[item])

    def map(self, fn):
        return _Dataset(self.examples, self.maps + (fn,))

    def filter(self, fn, lazy=True):
        return _Dataset([ex for ex in self.examples if fn(self._get(ex))], self.maps)

    def groupby(self, fn):
        out = {}
        for ex in self.examples:
            out.setdefault(fn(self._get(ex)), []).append(ex)
        return {k: _Dataset(v, self.maps) for k, v in out.items()}


def _concatenate(*datasets):
    return _Dataset([ex for d in datasets for ex in d.examples])


class _SoundFile:
    """Read-only subset of soundfile.SoundFile used by the reference's load_audio."""

    def __init__(self, path, mode='r', **kw):
        assert mode == 'r', mode
        self._w = wave.open(str(path), 'rb')
        assert self._w.getsampwidth() == 2
        self.samplerate = self._w.getframerate()
        self.channels = self._w.getnchannels()
        self.subtype = 'PCM_16'

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self._w.close()

    def __len__(self):
        return self._w.getnframes()

    def _prepare_read(self, start, stop, frames):
        total = len(self)
        start, stop, _ = slice(start, stop).indices(total)
        if stop < start:
            stop = start
        if frames < 0:
            frames = stop - start
        self._w.setpos(start)
        return frames

    def read(self, frames=-1, dtype='float64', fill_value=None, **kw):
        raw = self._w.readframes(frames)
        data = np.frombuffer(raw, dtype='<i2').astype(np.float64) / 2 ** 15
        if self.channels > 1:
            data = data.reshape(-1, self.channels)
        return data.astype(dtype)


def _tree(x):
    if isinstance(x, dict):
        return {k: _tree(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_tree(v) for v in x]
    return int(x)


def corpus_digest(root):
    h = hashlib.sha256()
    for p in sorted(Path(root).rglob('*.wav')):
        h.update(p.name.encode())
        h.update(p.read_bytes())
    return h.hexdigest()


def run_dev_sessions(core_module, json_path, name):
    """``session_id=dev``: both sessions through ONE Enhancer, as scripts/run.py:45-71 +
    core.py:333-394 do it."""
    import pb_chime5.mapping as ref_mapping
    sessions = DEV_CORPUS['session_id']
    n_total = int(DEV_CORPUS['seconds'] * 16000)
    for key in list(ref_mapping.session_array_to_num_samples):
        if key.split('_')[0] in sessions:
            ref_mapping.session_array_to_num_samples[key] = n_total
    enh = core_module.get_enhancer(database_path=str(json_path), **DEV_ENHANCER)
    it = enh.get_iterator(sessions)
    keys = ('start', 'end', 'num_samples', 'start_orig', 'end_orig', 'num_samples_orig')
    examples = [{'example_id': ex['example_id'], 'speaker_id': ex['speaker_id'],
                 'session_id': ex['session_id'], 'reference_array': ex['reference_array'],
                 **{k: _tree(ex[k]) for k in keys}} for ex in it]
    act = {}
    for session_id in sessions:
        act[session_id] = {
            array: {spk: [list(map(int, iv)) for iv in track.normalized_intervals]
                    for spk, track in tracks.items()}
            for array, tracks in enh.activity[session_id].items()}
    out = {}
    for idx in DEV_EXAMPLES:
        ex = it[idx]
        x_hat = enh.enhance_example(ex, debug=True)
        loc = enh.enhance_example_locals
        out[f'x_hat/{idx}'] = x_hat
        out[f'obs_shape/{idx}'] = np.array(loc['obs'].shape)
    fixture = {'corpus': {**DEV_CORPUS, 'chime6': False}, 'enhancer': DEV_ENHANCER,
               'corpus_sha256': corpus_digest(Path(json_path).parent),
               'examples': examples, 'activity': act, 'enhanced': list(DEV_EXAMPLES)}
    (HERE / f'{name}.json').write_text(json.dumps(fixture, indent=1))
    mg._save(f'{name}.npz', **out)
    print(f'{name}.json', len(examples), 'examples of', sessions,
          [tuple(out[f'obs_shape/{i}']) for i in DEV_EXAMPLES])


def run_front_door(core_module, json_path, tmp, name, chime6):
    import pb_chime5.mapping as ref_mapping
    session_id = CORPUS['session_id']
    n_total = int(CORPUS['seconds'] * 16000)
    for key in list(ref_mapping.session_array_to_num_samples):
        if key.startswith(session_id + '_'):
            ref_mapping.session_array_to_num_samples[key] = n_total

    enh = core_module.get_enhancer(database_path=str(json_path), **ENHANCER)
    it = enh.get_iterator(session_id)
    keys = ('start', 'end', 'num_samples', 'start_orig', 'end_orig', 'num_samples_orig')
    examples = [{'example_id': ex['example_id'], 'speaker_id': ex['speaker_id'],
                 'reference_array': ex['reference_array'],
                 **{k: _tree(ex[k]) for k in keys}} for ex in it]

    activity = enh.activity[session_id]
    if chime6:
        # 10 h tracks; only the intervals matter
        act = {spk: [list(map(int, iv)) for iv in track.normalized_intervals]
               for spk, track in activity.items()}
    else:
        act = {array: {spk: [list(map(int, iv)) for iv in track.normalized_intervals]
                       for spk, track in tracks.items()}
               for array, tracks in activity.items()}

    out = {}
    for idx in EXAMPLES:
        ex = it[idx]
        x_hat = enh.enhance_example(ex, debug=True)
        loc = enh.enhance_example_locals
        out[f'x_hat/{idx}'] = x_hat
        out[f'obs_shape/{idx}'] = np.array(loc['obs'].shape)
        out[f'activity/{idx}'] = np.packbits(
            np.array(list(loc['ex_array_activity'].values())), axis=-1)
        out[f'activity_len/{idx}'] = np.array(
            [len(v) for v in loc['ex_array_activity'].values()])

    fixture = {'corpus': {**CORPUS, 'chime6': chime6}, 'enhancer': ENHANCER,
               'corpus_sha256': corpus_digest(Path(json_path).parent),
               'examples': examples, 'activity': act, 'enhanced': list(EXAMPLES)}
    (HERE / f'{name}.json').write_text(json.dumps(fixture, indent=1))
    mg._save(f'{name}.npz', **out)
    print(f'{name}.json', len(examples), 'examples')


def main():
    from pb_chime5_amd.synthetic_corpus import write_chime5_corpus
    with tempfile.TemporaryDirectory() as tmp:
        tmp = Path(tmp)
        json5 = write_chime5_corpus(tmp / 'corpus5', **CORPUS)
        json6 = write_chime5_corpus(tmp / 'corpus6', **CORPUS, chime6=True)
        json_dev = write_chime5_corpus(tmp / 'corpus_dev', **DEV_CORPUS)
        ref = mg._prepare_reference(tmp)
        mg._register_stubs()
        mg._module('lazy_dataset', from_dict=lambda d: _Dataset(d.values()),
                   concatenate=_concatenate)
        mg._module('soundfile', SoundFile=_SoundFile)
        sys.path.insert(0, str(ref))
        pkg = types.ModuleType('pb_chime5')
        pkg.__path__ = [str(ref / 'pb_chime5')]
        pkg.git_root = ref
        sys.modules['pb_chime5'] = pkg
        import pb_chime5.core as core
        import pb_chime5.core_chime6 as core_chime6
        run_front_door(core, json5, tmp, 'chime5_session', chime6=False)
        run_front_door(core_chime6, json6, tmp, 'chime6_session', chime6=True)
        run_dev_sessions(core, json_dev, 'chime5_dev_sessions')


if __name__ == '__main__':
    main()
