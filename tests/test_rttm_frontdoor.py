"""RTTM-driven front door (core_chime6_rttm): integer / string logic on the CPU, and
one end-to-end session on the GPU checked against the oracle."""
import json

import numpy as np
import pytest

from conftest import rel_err

RTTM = """SPEAKER S02_U06.ENH 1 0.50 1.25 <NA> <NA> P05 <NA>
SPEAKER S02_U06.ENH 1 2.00 0.75 <NA> <NA> P05 <NA>
SPEAKER S02_U06.ENH 1 1.00 1.50 <NA> <NA> P06 <NA>
SPEAKER S09_U06 1 0.00 0.10 <NA> <NA> P25 <NA>
"""


def _make_chime6_dir(tmp_path, num_samples=52000, arrays=('U01', 'U02', 'U03')):
    from pb_chime5_amd.io import dump_audio, load_audio
    rng = np.random.default_rng(7)
    root = tmp_path / 'CHiME6'
    audio = {}
    (root / 'transcriptions' / 'dev').mkdir(parents=True)
    (root / 'transcriptions' / 'dev' / 'S02.json').write_text('[]')
    for a_i, a in enumerate(arrays):
        for ch in range(1, 5):
            n = num_samples - 37 * a_i            # files of different length
            x = rng.standard_normal(n) * 0.05
            path = root / 'audio' / 'dev' / f'S02_{a}.CH{ch}.wav'
            dump_audio(x, path, normalize=False)
            audio[path.name] = load_audio(path)
    rttm = tmp_path / 'dev_rttm'
    rttm.write_text(RTTM)
    return root, rttm, audio


def test_from_rttm_is_sample_exact(tmp_path):
    from pb_chime5_amd.database.chime5 import rttm
    p = tmp_path / 'x.rttm'
    p.write_text(RTTM)
    data = rttm.strip_file_id(rttm.from_rttm(p))
    assert sorted(data) == ['S02', 'S09']
    assert data['S02']['P05'].normalized_intervals == ((8000, 28000), (32000, 44000))
    assert data['S02']['P06'].normalized_intervals == ((16000, 40000),)
    assert data['S02']['P05'][7999:8002].tolist() == [False, True, True]
    # reference doctest (utils/intervall_array.py:45-58)
    q = tmp_path / 'd.rttm'
    q.write_text('SPEAKER S02 1 0 1 <NA> <NA> 1 <NA>\nSPEAKER S02 1 2 1 <NA> <NA> 1 <NA>\n'
                 'SPEAKER S02 1 0 2 <NA> <NA> 2 <NA>')
    d = rttm.from_rttm(q)
    assert repr(d['S02']['1']) == 'ArrayIntervall("0:16000, 32000:48000", shape=None)'
    assert repr(d['S02']['2']) == 'ArrayIntervall("0:32000", shape=None)'
    assert rttm.ones()[5:9].tolist() == [True] * 4 and rttm.zeros()[5:9].tolist() == [False] * 4
    # non-integer sample positions are refused like in the reference
    bad = tmp_path / 'bad.rttm'
    bad.write_text('SPEAKER S02 1 0.00001 1 <NA> <NA> 1 <NA>')
    with pytest.raises(AssertionError):
        rttm.from_rttm(bad)


def test_database_examples_context_and_audio(tmp_path):
    from pb_chime5_amd.database.chime5 import rttm
    from pb_chime5_amd.core_chime6_rttm import get_database
    root, rttm_file, audio = _make_chime6_dir(tmp_path)
    assert rttm.RTTMDatabase.example_id('S02', '1', 100, 200) == 'S02_U06.-1-000000100_000000200'
    files = rttm.get_chime6_files(root, worn=False, flat=False)
    assert list(files) == ['S02'] and list(files['S02']) == ['U01', 'U02', 'U03']
    assert [len(v) for v in files['S02'].values()] == [4, 4, 4]
    assert len(rttm.select_channels(root, True)['S02']) == 12
    outer = rttm.select_channels(root, 'outer_array_mics')['S02']
    assert [p.split('/')[-1] for p in outer] == [
        'S02_U01.CH1.wav', 'S02_U01.CH4.wav', 'S02_U02.CH1.wav', 'S02_U02.CH4.wav',
        'S02_U03.CH1.wav', 'S02_U03.CH4.wav']
    assert len(rttm.select_channels(root, 'first_array_mics')['S02']) == 3
    with pytest.raises(ValueError):
        rttm.select_channels(root, 'nope')

    db = get_database(root, rttm_file, 'outer_array_mics')
    assert 'dev' in db.dataset_names
    ds = db.get_dataset_for_session('dev', audio_read=True, context_samples=4000)
    ids = [e['example_id'] for e in ds.examples]
    assert ids == ['S02_U06.-P05-000008000_000028000', 'S02_U06.-P05-000032000_000044000',
                   'S02_U06.-P06-000016000_000040000']
    ex = ds[1]
    assert (ex['start'], ex['end'], ex['start_orig'], ex['end_orig'], ex['num_samples_orig']) == \
        (28000, 48000, 32000, 44000, 12000)
    assert ex['audio_data'].shape == (6, 20000) and ex['audio_data'].dtype == np.float64
    assert np.array_equal(ex['audio_data'][1], audio['S02_U01.CH4.wav'][28000:48000])
    # context is clipped at the start of the recording, files are cut to the shortest
    ds = db.get_dataset_for_session('S02', audio_read=True, context_samples=(10000, 9000))
    ex = ds[0]
    assert (ex['start'], ex['end']) == (0, 37000) and ex['start_orig'] == 8000
    ex = ds[1]
    assert ex['end'] == 53000 and ex['audio_data'].shape == (6, 52000 - 74 - 22000)


@pytest.mark.gpu
def test_enhance_session_end_to_end_vs_oracle(gpu_ctx, tmp_path):
    import gss_oracle as oracle
    from pb_chime5_amd.io import load_audio
    from pb_chime5_amd.scripts import enhance_rttm
    root, rttm_file, _ = _make_chime6_dir(tmp_path)
    out = tmp_path / 'out'
    enhance_rttm.main([
        '--chime6-dir', str(root), '--database-rttm', str(rttm_file), '--session-id', 'S02',
        '--out', str(out), '--context-samples', '4000', '--wpe-tabs', '2',
        '--bss-iterations', '3', '--multiarray', 'first_array_mics'])
    wavs = sorted((out / 'audio' / 'dev').glob('*.wav'))
    assert [w.name for w in wavs] == [
        'S02_U06.-P05-000008000_000028000.wav', 'S02_U06.-P05-000032000_000044000.wav',
        'S02_U06.-P06-000016000_000040000.wav']
    # recompute example 2 with the oracle from the same inputs
    from pb_chime5_amd.core_chime6_rttm import get_enhancer
    enh = get_enhancer(database_rttm=[str(rttm_file)], activity_rttm=[str(rttm_file)],
                       chime6_dir=root, multiarray='first_array_mics', context_samples=4000,
                       wpe_tabs=2, bss_iterations=3)
    ex = enh.get_dataset('S02')[1]
    act = {k: v[ex['start']:ex['end']] for k, v in enh.activity['S02'].items()}
    assert list(act) == ['P05', 'P06', 'Noise'] and act['Noise'].all()
    oex = {'start': {'original': ex['start']}, 'start_orig': {'original': ex['start_orig']},
           'end_orig': {'original': ex['end_orig']}, 'end': {'original': ex['end']}}
    want = oracle.enhance_observation(ex['audio_data'], np.array(list(act.values())), 0, oex,
                                      wpe_taps=2, bss_iterations=3,
                                      gss_fn=oracle.gss_block_batched)
    want = want[4000:4000 + 12000]
    got = enh.enhance_example(ex)
    assert got.shape == want.shape == (12000,)
    assert rel_err(got, want) < 1e-6
    # the file holds the peak-normalised 16-bit version of the same signal
    pcm = load_audio(wavs[1])
    ref = want * ((2 ** 15 - 1) / 2 ** 15 / np.max(np.abs(want)))
    assert np.max(np.abs(pcm - ref)) <= 1.01 / 2 ** 15


@pytest.mark.gpu
def test_rttm_with_nine_speakers_runs_like_the_reference(gpu_ctx, tmp_path):
    """A diarisation RTTM with 9 speakers -> K = 10 classes with the garbage class
    (core_chime6_rttm.py:36-69 builds one activity row per RTTM speaker; pb_bss accepts
    K < 20).  Rounds 1-2 stopped at 8 classes."""
    import gss_oracle as oracle
    from pb_chime5_amd.core_chime6_rttm import get_enhancer
    root, rttm_file, _ = _make_chime6_dir(tmp_path)
    lines = []
    for k in range(9):
        lines.append(f'SPEAKER S02_U06.ENH 1 {0.25 * k:.2f} {0.75 + 0.125 * (k % 3):.3f} <NA> <NA> '
                     f'spk{k} <NA>\n')
    rttm_file.write_text(''.join(lines))
    enh = get_enhancer(database_rttm=[str(rttm_file)], activity_rttm=[str(rttm_file)],
                       chime6_dir=root, multiarray='outer_array_mics', context_samples=8000,
                       wpe_tabs=2, bss_iterations=3)
    ds = enh.get_dataset('S02')
    assert len(ds) == 9
    ex = ds[4]
    act = {k: v[ex['start']:ex['end']] for k, v in enh.activity['S02'].items()}
    assert len(act) == 10 and list(act)[-1] == 'Noise'
    target = list(act).index(ex['speaker_id'])
    oex = {'start': {'original': ex['start']}, 'start_orig': {'original': ex['start_orig']},
           'end_orig': {'original': ex['end_orig']}, 'end': {'original': ex['end']}}
    want = oracle.enhance_observation(ex['audio_data'], np.array(list(act.values())), target, oex,
                                      wpe_taps=2, bss_iterations=3,
                                      gss_fn=oracle.gss_block_batched)
    lo = ex['start_orig'] - ex['start']
    want = want[lo:lo + ex['num_samples_orig']]
    got = enh.enhance_example(ex)
    assert got.shape == want.shape
    assert rel_err(got, want) < 1e-5


# ---------------------------------------------------------------- reference fixture
GOLDEN = __import__('pathlib').Path(__file__).parent / 'golden'


def test_one_example_per_rttm_line_in_file_order(tmp_path):
    """The reference walks the segments of a speaker as the RTTM lists them (rttm.py:466,
    ``speaker.intervals``: not merged, not sorted): two touching segments are two utterances, an
    earlier segment further down the file comes later, a repeated line is one example.  (Found by
    tests/golden/fuzz_rttm_vs_reference.py, which runs random RTTM files through the reference's
    own front door and this one: 120 files, identical examples and activity.)"""
    from pb_chime5_amd.core_chime6_rttm import get_enhancer
    root, rttm_file, _ = _make_chime6_dir(tmp_path)
    rttm_file.write_text(
        'SPEAKER S02_U06.ENH 1 0.50 0.50 <NA> <NA> P05 <NA>\n'
        'SPEAKER S02_U06.ENH 1 1.00 0.25 <NA> <NA> P05 <NA>\n'
        'SPEAKER S02_U06.ENH 1 0.70 0.10 <NA> <NA> P06 <NA>\n'
        'SPEAKER S02_U06.ENH 1 0.20 0.10 <NA> <NA> P05 <NA>\n'
        'SPEAKER S02_U06.ENH 1 1.00 0.25 <NA> <NA> P05 <NA>\n')
    enh = get_enhancer(database_rttm=str(rttm_file), activity_rttm=str(rttm_file),
                       chime6_dir=str(root), multiarray='first_array_mics', context_samples=0,
                       wpe=False, bss_iterations=1)
    got = [(ex['speaker_id'], ex['start'], ex['end']) for ex in enh.get_dataset('dev')]
    assert got == [('P05', 8000, 16000), ('P05', 16000, 20000), ('P05', 3200, 4800),
                   ('P06', 11200, 12800)]
    # the activity, on the other hand, is the merged one
    assert [tuple(map(int, iv)) for iv in enh.activity['S02']['P05'].normalized_intervals] == \
        [(3200, 4800), (8000, 20000)]


def _reference_setup(tmp_path):
    """The directory and RTTM that tests/golden/make_golden_rttm.py fed to the reference's
    own rttm.py / core_chime6_rttm.py (the RTTM restricted to the session that has audio:
    the reference builds examples for every session of the file)."""
    root, rttm_file, audio = _make_chime6_dir(tmp_path)
    rttm_file.write_text(''.join(l + '\n' for l in RTTM.splitlines() if ' S02' in l))
    return root, rttm_file


def _multiarray(tag):
    return True if tag == 'True' else tag


def test_examples_and_activity_match_reference(tmp_path):
    """Example enumeration, ids, context, channel selection, audio cut to the shortest
    file and the RTTM activity: bit exact against the reference's own code."""
    from pb_chime5_amd.core_chime6_rttm import get_enhancer
    from pb_chime5_amd.database.chime5 import rttm
    fx = json.loads((GOLDEN / 'rttm_session.json').read_text())
    root, rttm_file = _reference_setup(tmp_path)
    assert rttm.RTTMDatabase.example_id('S02', '1', 100, 200) == fx['example_id']
    for tag, case in fx['cases'].items():
        enh = get_enhancer(database_rttm=str(rttm_file), activity_rttm=str(rttm_file),
                           chime6_dir=str(root), multiarray=_multiarray(tag), **fx['enhancer'])
        assert sorted(enh.db.dataset_names) == fx['dataset_names']
        ds = enh.get_dataset('dev')
        assert len(ds) == len(case['examples'])
        for ex, want in zip(ds, case['examples']):
            for key in ('example_id', 'start', 'end', 'num_samples', 'session_id', 'speaker_id',
                        'dataset', 'start_orig', 'end_orig', 'num_samples_orig'):
                assert ex[key] == want[key], (tag, key)
            assert [p.split('/')[-1] for p in ex['audio_path']] == want['audio_files'], tag
            assert list(ex['audio_data'].shape) == want['audio_shape'], tag
        activity = enh.activity['S02']
        assert list(activity.keys()) == list(case['activity'].keys())
        for spk, want in case['activity'].items():
            if want == 'ones':
                assert activity[spk][5:50000].all()
            else:
                assert [list(map(int, iv)) for iv in activity[spk].normalized_intervals] == want


@pytest.mark.gpu
def test_enhance_example_matches_reference_fixture(gpu_ctx, tmp_path):
    from pb_chime5_amd.core_chime6_rttm import get_enhancer
    fx = json.loads((GOLDEN / 'rttm_session.json').read_text())
    gold = np.load(GOLDEN / 'rttm_session.npz')
    root, rttm_file = _reference_setup(tmp_path)
    checked = 0
    for tag in fx['cases']:
        enh = get_enhancer(database_rttm=str(rttm_file), activity_rttm=str(rttm_file),
                           chime6_dir=str(root), multiarray=_multiarray(tag), **fx['enhancer'])
        ds = enh.get_dataset('dev')
        for i in range(len(ds)):
            key = f'{tag}/x_hat/{i}'
            if key not in gold:
                continue
            got = enh.enhance_example(ds[i])
            assert got.shape == gold[key].shape, key
            assert rel_err(got, gold[key]) < 1e-4, key      # north-star tolerance
            checked += 1
    assert checked == 5
