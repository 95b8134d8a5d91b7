"""A small synthetic corpus in the CHiME-5 layout: per-array, per-channel PCM16 WAV
files plus the example JSON the reference's ``create_json`` would write
(/root/reference/pb_chime5/database/chime5/create_json.py:306-475: ``datasets`` keyed
by session id, ``alias`` dev -> sessions; per example ``audio_path`` / ``start`` /
``end`` / ``num_samples`` nested by 'observation' array and 'worn' microphone plus the
'original' clock, ``speaker_id``, ``session_id``, ``transcription``,
``reference_array``, ``location``).

There is no CHiME-5 data in the build or test environment; this lets the whole
session driver -- JSON database, annotation activity, context bookkeeping, WAV
reading, enhancement, WAV writing -- run end to end.  Each array has its own clock:
utterance boundaries are shifted by a per-array offset and a per-utterance jitter and
differ slightly in duration, as in the real corpus, so that ``adjust_start_end`` and
``AddContext(equal_start_context=True)`` have something to do.
"""
import json
from pathlib import Path

import numpy as np

from pb_chime5_amd import mapping
from pb_chime5_amd.io import dump_audio
from pb_chime5_amd.synthetic import SAMPLE_RATE, _rir, _source


def _example_id(speaker_id, session_id, start, end, chime6=False):
    a = str(int(start * 100 / SAMPLE_RATE)).zfill(7)
    b = str(int(end * 100 / SAMPLE_RATE)).zfill(7)
    if chime6:
        return f'{speaker_id}_{session_id}-{a}-{b}'
    return f'{speaker_id}_{session_id}_{a}-{b}'


def write_chime5_corpus(root, session_id='S02', seconds=10.0, seed=11, utts_per_speaker=2,
                        num_redacted=1, rir_taps=256, chime6=False, noise=1e-3):
    """Writes ``root/audio/<dataset>/<session>_<array>.CH<m>.wav`` and
    ``root/chime5.json``; returns the path of the JSON.  ``chime6=True`` writes the
    CHiME-6 flavour instead (``root/chime6.json``): one synchronised clock, so start /
    end / num_samples are plain integers (create_json.py:361-363,436-439).
    ``session_id`` may be a list (``['S02', 'S09']`` = the dev set: S09 has five arrays,
    mapping.py:67): every session gets its own audio (seed + its position in the list) and
    ONE database holds them all, ``alias`` dev -> both, as create_json writes it.
    ``noise``: sensor noise relative to the sources."""
    if not isinstance(session_id, str):
        sessions = {}
        for i, sid in enumerate(session_id):
            sessions[sid] = _write_chime5_session(root, sid, seconds, seed + i, utts_per_speaker,
                                                  num_redacted, rir_taps, chime6, noise)
        return _write_database(root, sessions, chime6)
    return _write_database(root, {session_id: _write_chime5_session(
        root, session_id, seconds, seed, utts_per_speaker, num_redacted, rir_taps, chime6,
        noise)}, chime6)


def _write_chime5_session(root, session_id, seconds, seed, utts_per_speaker, num_redacted,
                          rir_taps, chime6, noise):
    from scipy.signal import fftconvolve
    root = Path(root)
    rng = np.random.default_rng(seed)
    dataset = mapping.session_to_dataset[session_id]
    speakers = mapping.session_to_speakers[session_id]
    arrays = mapping.session_to_arrays[session_id]
    n_total = int(seconds * SAMPLE_RATE)
    audio_dir = root / 'audio' / dataset
    audio_dir.mkdir(parents=True, exist_ok=True)

    # ---- utterances on the 'original' clock
    utterances = []
    for spk in speakers:
        for _ in range(utts_per_speaker):
            length = int(rng.uniform(0.8, 2.0) * SAMPLE_RATE)
            start = int(rng.integers(SAMPLE_RATE // 2, n_total - length - SAMPLE_RATE // 2))
            utterances.append((spk, start, start + length, 'some words'))
    for _ in range(num_redacted):
        spk = speakers[int(rng.integers(len(speakers)))]
        start = int(rng.integers(0, n_total - SAMPLE_RATE))
        utterances.append((spk, start, start + SAMPLE_RATE // 2, '[redacted]'))
    utterances.sort(key=lambda u: (u[1], u[0]))

    # ---- audio: every speaker talks inside their (non redacted) utterances
    channels = [(a, m) for a in arrays for m in range(1, 5)]
    obs = np.zeros((len(channels), n_total))
    for spk in speakers:
        act = np.zeros(n_total, dtype=bool)
        for s, a, b, words in utterances:
            if s == spk and words != '[redacted]':
                act[a:b] = True
        src = _source(rng, n_total) * act
        for d in range(len(channels)):
            obs[d] += fftconvolve(src, _rir(rng, rir_taps))[:n_total]
    # (sensor noise 60 dB down by default; with all 24 microphones the noise PSD matrix of a few
    # point sources is then singular to rounding and the reference's own float64 beamformer
    # output is rounding noise -- fixtures that compare beamformed signals at 24 channels ask for
    # more noise)
    obs += rng.standard_normal(obs.shape) * noise
    obs *= 0.05
    for d, (a, m) in enumerate(channels):
        dump_audio(obs[d], audio_dir / f'{session_id}_{a}.CH{m}.wav', normalize=False)

    return _session_examples(rng, session_id, utterances, n_total, audio_dir, chime6)


def _session_examples(rng, session_id, utterances, n_total, audio_dir, chime6):
    """The examples of one session for `utterances` [(speaker, start, end, words) on the
    'original' clock]: per-array / per-worn-microphone clocks with an offset and a
    per-utterance jitter."""
    speakers = mapping.session_to_speakers[session_id]
    arrays = mapping.session_to_arrays[session_id]
    audio_path = {
        'observation': {a: [str(audio_dir / f'{session_id}_{a}.CH{m}.wav') for m in range(1, 5)]
                        for a in arrays},
        'worn': {p: str(audio_dir / f'{session_id}_{p}.wav') for p in speakers},
    }

    # ---- per-clock boundaries
    array_offset = {a: int(rng.integers(-300, 301)) for a in arrays}
    worn_offset = {p: int(rng.integers(-100, 101)) for p in speakers}
    examples = {}
    for spk, start, end, words in utterances:
        def clock(offset, jitter):
            s = int(np.clip(start + offset + rng.integers(-jitter, jitter + 1), 0, n_total - 2))
            e = int(np.clip(end + offset + rng.integers(-jitter, jitter + 1), s + 1, n_total))
            return s, e
        obs_times = {a: clock(array_offset[a], 40) for a in arrays}
        worn_times = {p: clock(worn_offset[p], 10) for p in speakers}
        start_d = {'observation': {a: t[0] for a, t in obs_times.items()},
                   'worn': {p: t[0] for p, t in worn_times.items()}, 'original': start}
        end_d = {'observation': {a: t[1] for a, t in obs_times.items()},
                 'worn': {p: t[1] for p, t in worn_times.items()}, 'original': end}
        num_d = {'observation': {a: t[1] - t[0] for a, t in obs_times.items()},
                 'worn': {p: t[1] - t[0] for p, t in worn_times.items()},
                 'original': end - start}
        if chime6:
            start_d, end_d, num_d = start, end, end - start
        ex = {
            'session_id': session_id,
            'num_samples': num_d,
            'audio_path': audio_path,
            'notes': [],
            'start': start_d,
            'end': end_d,
            'transcription': words,
            'speaker_id': spk,
            'gender': 'male',
            'location': 'kitchen',
            'reference_array': arrays[int(rng.integers(len(arrays)))],
        }
        examples[_example_id(spk, session_id, start, end, chime6)] = ex

    return examples


def _write_database(root, sessions, chime6):
    """`sessions` {session id: examples} -> root/chime5.json (chime6.json): ``datasets`` keyed
    by session, ``alias`` dataset -> its sessions (create_json.py:306-475)."""
    alias = {}
    for session_id in sessions:
        alias.setdefault(mapping.session_to_dataset[session_id], []).append(session_id)
    database = {'datasets': dict(sessions), 'alias': alias}
    json_path = Path(root) / ('chime6.json' if chime6 else 'chime5.json')
    with open(json_path, 'w') as fd:
        json.dump(database, fd, indent=1, sort_keys=True)
    return json_path


def _write_json(root, rng, session_id, utterances, n_total, audio_dir, chime6):
    return _write_database(root, {session_id: _session_examples(
        rng, session_id, utterances, n_total, audio_dir, chime6)}, chime6)


def write_dev_shaped_corpus(root, session_ids=('S02', 'S09'), seconds=660.0, num_utterances=220,
                            seed=4, rir_taps=512):
    """`write_dev_shaped_session` for several sessions behind ONE database: ``session_id=dev``
    of the reference's scripts is S02 (6 arrays = 24 channels) AND S09 (5 arrays = 20 channels,
    mapping.py:67) -- the channel count changes in the middle of a run.  `num_utterances` and
    `seconds` per session."""
    sessions = {}
    for i, sid in enumerate(session_ids):
        sessions[sid] = write_dev_shaped_session(root, sid, seconds, num_utterances, seed + i,
                                                 rir_taps, database=False)
    return _write_database(root, sessions, False)


def write_dev_shaped_session(root, session_id='S02', seconds=660.0, num_utterances=220, seed=4,
                             rir_taps=512, block=1 << 18, database=True):
    """A session at the size the CHiME-5 dev set has per utterance (BASELINE.json configs[3]
    stand-in; the corpus itself is not available): 6 arrays x 4 per-channel PCM16 files of
    `seconds` of audio, `num_utterances` utterances whose lengths follow the dev-shaped draw of
    synthetic.config3_core_samples (~ LogNormal(ln 2.5 s, 0.7) clipped to [0.5 s, 15 s]),
    speakers talking over each other as at a dinner party.  Same JSON layout as
    write_chime5_corpus.  The audio is synthesised block-wise (overlap-add of the reverberated
    sources, one batched FFT per block) straight into int16, so that ten minutes x 24 channels
    take seconds and ~0.5 GB instead of minutes and several GB."""
    import scipy.fft as sfft
    from pb_chime5_amd.synthetic import config3_core_samples
    root = Path(root)
    rng = np.random.default_rng(seed)
    dataset = mapping.session_to_dataset[session_id]
    speakers = mapping.session_to_speakers[session_id]
    arrays = mapping.session_to_arrays[session_id]
    n_total = int(seconds * SAMPLE_RATE)
    audio_dir = root / 'audio' / dataset
    audio_dir.mkdir(parents=True, exist_ok=True)

    # utterances stay one context (15 s) + 1 s away from both ends of the recording when it is
    # long enough: an utterance whose nominal context reaches past the recording has its whole
    # target mask zeroed by bf_drop_context and ends in 0 / 0 -- in the reference too
    margin = 16 * SAMPLE_RATE if n_total > 80 * SAMPLE_RATE else SAMPLE_RATE // 2
    utterances = []
    for i in range(num_utterances):
        spk = speakers[i % len(speakers)]
        length = min(config3_core_samples(seed * 100000 + i), n_total // 4)
        start = int(rng.integers(margin, n_total - length - margin))
        utterances.append((spk, start, start + length, 'some words'))
    utterances.sort(key=lambda u: (u[1], u[0]))

    channels = [(a, m) for a in arrays for m in range(1, 5)]
    D, S = len(channels), len(speakers)
    acts = np.zeros((S, n_total), dtype=bool)
    for spk, a, b, _ in utterances:
        acts[speakers.index(spk), a:b] = True
    sources = np.stack([_source(rng, n_total) for _ in range(S)]) * acts
    nfft = sfft.next_fast_len(block + rir_taps - 1, real=True)
    H = sfft.rfft(np.stack([[_rir(rng, rir_taps) for _ in range(D)] for _ in range(S)]),
                  nfft, axis=-1, workers=-1)                                   # (S, D, nf)
    pcm = np.empty((D, n_total), dtype=np.int16)
    tail = np.zeros((D, rir_taps - 1))
    for b0 in range(0, n_total, block):
        nb = min(block, n_total - b0)
        X = sfft.rfft(sources[:, b0:b0 + nb], nfft, axis=-1, workers=-1)        # (S, nf)
        y = sfft.irfft(np.einsum('sf,sdf->df', X, H), nfft, axis=-1, workers=-1)
        y[:, :rir_taps - 1] += tail
        tail = y[:, nb:nb + rir_taps - 1].copy()
        y = y[:, :nb] + rng.standard_normal((D, nb)) * 3e-2                    # sensor noise
        pcm[:, b0:b0 + nb] = np.clip(np.rint(y * (0.03 * 32768)), -32768, 32767)
    for d, (a, m) in enumerate(channels):
        dump_audio(pcm[d], audio_dir / f'{session_id}_{a}.CH{m}.wav', normalize=False)
    examples = _session_examples(rng, session_id, utterances, n_total, audio_dir, False)
    return _write_database(root, {session_id: examples}, False) if database else examples
