"""CHiME-5 JSON front door: JSON database -> example
iterator with context bookkeeping, annotation activity, WAV reading, session driver.

Expected values come from tests/golden/chime5_session.{json,npz}, written by
tests/golden/make_golden_session.py from the reference's real session code
(database.py:83-131, activity.py:8-222, core.py:333-512, io/audioread.py) on the
synthetic corpus that the tests regenerate here bit-identically."""
import hashlib
import json
from pathlib import Path

import numpy as np
import pytest

GOLDEN = Path(__file__).parent / 'golden'


def _load_fixture(name):
    fx = json.loads((GOLDEN / f'{name}.json').read_text())
    fx['name'] = name
    return fx


def _write_corpus(fx, root):
    from pb_chime5_amd.synthetic_corpus import write_chime5_corpus
    json_path = write_chime5_corpus(root, **fx['corpus'])
    h = hashlib.sha256()
    for p in sorted(root.rglob('*.wav')):
        h.update(p.name.encode())
        h.update(p.read_bytes())
    assert h.hexdigest() == fx['corpus_sha256'], 'synthetic corpus is not reproducible here'
    return json_path


@pytest.fixture(scope='module')
def fixture():
    return _load_fixture('chime5_session')


@pytest.fixture(scope='module')
def corpus(fixture, tmp_path_factory):
    return _write_corpus(fixture, tmp_path_factory.mktemp('chime5_corpus'))


@pytest.fixture(scope='module')
def fixture6():
    return _load_fixture('chime6_session')


@pytest.fixture(scope='module')
def corpus6(fixture6, tmp_path_factory):
    return _write_corpus(fixture6, tmp_path_factory.mktemp('chime6_corpus'))


def _enhancer(corpus, fixture, **kw):
    if fixture['corpus'].get('chime6'):
        from pb_chime5_amd.core_chime6 import get_enhancer
    else:
        from pb_chime5_amd.core import get_enhancer
    return get_enhancer(database_path=str(corpus), **{**fixture['enhancer'], **kw})


def test_iterator_bookkeeping_matches_reference(corpus, fixture):
    enh = _enhancer(corpus, fixture)
    it = enh.get_iterator(fixture['corpus']['session_id'])
    assert len(it) == len(fixture['examples'])
    for ex, want in zip(it, fixture['examples']):
        assert ex['example_id'] == want['example_id']
        assert ex['speaker_id'] == want['speaker_id']
        assert ex['reference_array'] == want['reference_array']
        for key in ('start', 'end', 'num_samples', 'start_orig', 'end_orig', 'num_samples_orig'):
            assert ex[key] == want[key], (ex['example_id'], key)
        assert ex['transcription'] != '[redacted]'
    # the redacted utterance is in the database but not in the iterator
    raw = enh.db.get_datasets(fixture['corpus']['session_id'])
    assert len(raw) == len(it) + fixture['corpus']['num_redacted']


def test_annotation_activity_matches_reference(corpus, fixture):
    enh = _enhancer(corpus, fixture)
    session_id = fixture['corpus']['session_id']
    activity = enh.activity[session_id]
    assert list(activity.keys()) == list(fixture['activity'].keys())
    for array, tracks in fixture['activity'].items():
        assert list(activity[array].keys()) == list(tracks.keys())
        for spk, intervals in tracks.items():
            got = [list(map(int, iv)) for iv in activity[array][spk].normalized_intervals]
            assert got == intervals, (array, spk)
    assert enh.activity[session_id] is activity          # one session is cached


def test_chime6_iterator_and_activity_match_reference(corpus6, fixture6):
    """core_chime6.py:322-331 (no time adjustment, plain context) and activity.py:225-403."""
    enh = _enhancer(corpus6, fixture6)
    it = enh.get_iterator('S02')
    assert len(it) == len(fixture6['examples'])
    for ex, want in zip(it, fixture6['examples']):
        assert ex['example_id'] == want['example_id']
        for key in ('start', 'end', 'num_samples', 'start_orig', 'end_orig', 'num_samples_orig'):
            assert isinstance(ex[key], int) and ex[key] == want[key], (ex['example_id'], key)
    activity = enh.activity['S02']
    assert list(activity.keys()) == list(fixture6['activity'].keys())
    for spk, intervals in fixture6['activity'].items():
        assert [list(map(int, iv)) for iv in activity[spk].normalized_intervals] == intervals


def test_get_activity_perspectives_and_dense(corpus, fixture):
    from pb_chime5_amd.activity import get_activity
    from pb_chime5_amd.database.chime5.database import Chime5
    db = Chime5(corpus)
    it = db.get_datasets('S02')
    n_total = int(fixture['corpus']['seconds'] * 16000)
    dense = get_activity(it, perspective='array', garbage_class=None, use_ArrayIntervall=False)
    sparse = get_activity(it, perspective='array', garbage_class=True, use_ArrayIntervall=True)
    assert 'Noise' not in dense['S02']['U01'] and 'Noise' in sparse['S02']['U01']
    for array in dense['S02']:
        for spk, track in dense['S02'][array].items():
            assert track.dtype == bool and track.shape == (n_total,)
            np.testing.assert_array_equal(track, sparse['S02'][array][spk][:])
    worn = get_activity(it, perspective='global_worn', garbage_class=2, use_ArrayIntervall=True,
                        num_samples={'S02_P': n_total})
    assert list(worn['S02'].keys()) == ['P']
    assert list(worn['S02']['P'].keys()) == ['P05', 'P06', 'P07', 'P08', 'Noise0', 'Noise1']
    one = get_activity(it, perspective='U03', garbage_class=False, use_ArrayIntervall=True)
    assert list(one['S02'].keys()) == ['U03']
    assert not one['S02']['U03']['Noise'][:].any()
    with pytest.raises(ValueError):
        get_activity(it, perspective='array', garbage_class='yes')


def test_example_list_operations():
    from pb_chime5_amd.database import DictDatabase
    db = DictDatabase({'datasets': {'A': {'a1': {'v': 1}, 'a2': {'v': 2}}, 'B': {'b1': {'v': 3}}},
                       'alias': {'all': ['A', 'B']}})
    it = db.get_datasets('all')
    assert len(it) == 3 and it.keys() == ('a1', 'a2', 'b1')
    assert it[0]['dataset'] == 'all' and db.get_datasets('A')[0]['dataset'] == 'A'

    def bump(ex):
        ex['v'] += 10
        return ex
    mapped = it.map(bump)
    assert [ex['v'] for ex in mapped] == [11, 12, 13]
    assert [ex['v'] for ex in mapped] == [11, 12, 13]          # examples are copied on access
    assert [ex['v'] for ex in mapped.filter(lambda ex: ex['v'] > 11, lazy=False)] == [12, 13]
    assert [ex['v'] for ex in mapped[1:]] == [12, 13]
    assert [ex['v'] for ex in mapped[slice(0, None, 2)]] == [11, 13]
    groups = mapped.groupby(lambda ex: ex['v'] % 2)
    assert sorted(groups) == [0, 1] and len(groups[1]) == 2
    assert mapped['b1']['v'] == 13
    with pytest.raises(KeyError):
        db.get_datasets('C')
    with pytest.raises(RuntimeError):
        DictDatabase({'datasets': {'E': {}}}).get_datasets('E')


def test_example_id_map_fn():
    from pb_chime5_amd.database.chime5.database import Chime5
    # the two doctest vectors of database.py:59-81
    assert Chime5.example_id_map_fn({'example_id': 'P05_S02_0004060-0004382', 'dataset': 'dev',
                                     'location': 'kitchen'}) == 'P05_S02_KITCHEN.L-0004060-0004382'
    assert Chime5.example_id_map_fn({'example_id': 'P09_S03_0005948-0006038', 'dataset': 'train',
                                     'location': 'unknown'}) == 'P09_S03_NOLOCATION.L-0005948-0006038'


def test_wav_known_answers(tmp_path):
    """The doctest vectors of io/audiowrite.py:37-64 (float -> PCM16 -> float)."""
    from pb_chime5_amd.io import dump_audio, load_audio
    f = tmp_path / 'x.wav'
    a = np.array([1, 2, -4, 4], dtype=np.int16)
    dump_audio(a, f, normalize=False)
    np.testing.assert_array_equal(load_audio(f) * 2 ** 15, [1., 2., -4., 4.])
    dump_audio(a, f, normalize=True)
    np.testing.assert_allclose(load_audio(f), [0.24996948, 0.49996948, -0.99996948, 0.99996948],
                               atol=5e-9)
    np.testing.assert_array_equal(load_audio(f) * 2 ** 15, [8191., 16383., -32767., 32767.])
    data = np.arange(10) / 32
    dump_audio(data, f, normalize=False)
    np.testing.assert_array_equal(load_audio(f), data)
    np.testing.assert_array_equal(load_audio(f, start=2, stop=5), data[2:5])
    np.testing.assert_array_equal(load_audio([f, f]).shape, (2, 10))


def test_loaded_observation_shape(corpus, fixture):
    """Channel selection / alignment of enhance_example without touching the GPU."""
    from pb_chime5_amd.io import load_audio
    enh = _enhancer(corpus, fixture)
    it = enh.get_iterator('S02')
    gold = np.load(GOLDEN / 'chime5_session.npz')
    for idx in fixture['enhanced']:
        ex = it[idx]
        arrays = [load_audio(ex['audio_path']['observation'][a], start=ex['start']['observation'][a],
                             stop=ex['end']['observation'][a])
                  for a in sorted(ex['audio_path']['observation'])]
        n = min(v.shape[-1] for v in arrays)
        assert (2 * len(arrays), n) == tuple(gold[f'obs_shape/{idx}'])


@pytest.mark.gpu
@pytest.mark.parametrize('flavour', ['chime5', 'chime6'])
def test_enhance_example_matches_reference(flavour, request):
    from tests.conftest import rel_err
    fixture = request.getfixturevalue('fixture' if flavour == 'chime5' else 'fixture6')
    corpus = request.getfixturevalue('corpus' if flavour == 'chime5' else 'corpus6')
    enh = _enhancer(corpus, fixture)
    it = enh.get_iterator('S02')
    gold = np.load(GOLDEN / f'{flavour}_session.npz')
    for idx in fixture['enhanced']:
        ex = it[idx]
        x_hat = enh.enhance_example(ex, debug=True)
        loc = enh.enhance_example_locals
        assert loc['obs'].shape == tuple(gold[f'obs_shape/{idx}'])
        act = np.array(list(loc['ex_array_activity'].values()))
        assert [len(v) for v in loc['ex_array_activity'].values()] == list(gold[f'activity_len/{idx}'])
        np.testing.assert_array_equal(np.packbits(act, axis=-1), gold[f'activity/{idx}'])
        want = gold[f'x_hat/{idx}']
        assert x_hat.shape == want.shape
        n_orig = ex['num_samples_orig']
        if flavour == 'chime5':
            n_orig = n_orig['observation'][ex['reference_array']]
        assert x_hat.shape[0] == n_orig
        # BASELINE.json north_star tolerance: 1e-4 relative (observed ~4e-6: five EM
        # iterations amplify the last-bit differences of the WPE solve)
        assert rel_err(x_hat, want) < 1e-4


@pytest.mark.gpu
def test_enhance_session_writes_reference_layout(corpus, fixture, tmp_path):
    """core.py:333-394: audio_dir/<dataset>/<example_id>.wav, peak normalised PCM16."""
    from pb_chime5_amd.io import load_audio
    enh = _enhancer(corpus, fixture)
    audio_dir = tmp_path / 'audio'
    enh.enhance_session('S02', audio_dir, dataset_slice=slice(0, None, 5))
    written = sorted(p.relative_to(audio_dir).as_posix() for p in audio_dir.rglob('*.wav'))
    ids = [ex['example_id'] for ex in fixture['examples']][0::5]
    assert written == sorted(f'dev/{i}.wav' for i in ids)
    assert (audio_dir / 'train').is_dir() and (audio_dir / 'eval').is_dir()
    gold = np.load(GOLDEN / 'chime5_session.npz')
    for idx in fixture['enhanced']:
        if idx % 5:
            continue
        got = load_audio(audio_dir / 'dev' / f"{fixture['examples'][idx]['example_id']}.wav")
        want = gold[f'x_hat/{idx}']
        want = want * ((2 ** 15 - 1) / 2 ** 15 / np.max(np.abs(want)))
        assert np.max(np.abs(got - want)) <= 2.0 / 2 ** 15       # floor to PCM16 + 1 LSB
    with pytest.raises(FileExistsError):
        enh.enhance_session('S02', audio_dir, dataset_slice=True)
    enh.enhance_session('S02', audio_dir, dataset_slice=True, audio_dir_exist_ok=True)
    assert len(list(audio_dir.rglob('*.wav'))) == 3       # examples 0, 5 and now 1


@pytest.mark.gpu
def test_pipelined_session_equals_sequential(corpus, fixture, tmp_path):
    """enhance_session keeps two utterances in flight; the files must be byte-identical
    to the one-at-a-time loop."""
    a, b = tmp_path / 'seq', tmp_path / 'pipe'
    seq = _enhancer(corpus, fixture)
    seq.inflight = 1
    seq.enhance_session('S02', a)
    pipe = _enhancer(corpus, fixture)
    assert pipe.inflight == 2
    pipe.enhance_session('S02', b)
    files = sorted(p.relative_to(a) for p in a.rglob('*.wav'))
    assert len(files) == len(fixture['examples'])
    assert files == sorted(p.relative_to(b) for p in b.rglob('*.wav'))
    for rel in files:
        assert (a / rel).read_bytes() == (b / rel).read_bytes(), rel
    # where the host thread of the pipelined session spent its time (bench.py --config 4s)
    clock = pipe.session_clock
    assert clock['examples'] == len(files) and clock['wall_s'] > 0
    assert clock['gpu_wait_s'] + clock['host_wait_s'] + clock['enqueue_s'] <= clock['wall_s']


@pytest.mark.gpu
@pytest.mark.parametrize('multiarray,loaders,inflight', [
    (True, 1, 2), (True, 4, 3), ('outer_array_mics', 3, 2), ('first_array_mics', 2, 2), (False, 3, 2)])
def test_loader_pool_session_is_byte_identical(corpus, fixture, tmp_path, multiarray, loaders,
                                               inflight):
    """The session driver's host side -- loader threads reading WAV slices straight into
    page-locked int16 rows, asynchronous H2D / D2H of the trimmed range, a writer thread --
    against the reference-shaped loop (load_audio -> float64 -> enhance_example -> dump_audio,
    core.py:363-392): every WAV byte-identical, for every channel selection and any number of
    loader threads / utterances in flight."""
    a, b = tmp_path / 'seq', tmp_path / 'pool'
    kw = dict(multiarray=multiarray, wpe_tabs=2, bss_iterations=3)
    seq = _enhancer(corpus, fixture, **kw)
    seq.inflight = 1
    seq.enhance_session('S02', a, dataset_slice=slice(0, 7))
    pool = _enhancer(corpus, fixture, **kw)
    pool.inflight, pool.loaders = inflight, loaders
    pool.enhance_session('S02', b, dataset_slice=slice(0, 7))
    files = sorted(p.relative_to(a) for p in a.rglob('*.wav'))
    assert len(files) == 7 and files == sorted(p.relative_to(b) for p in b.rglob('*.wav'))
    for rel in files:
        assert (a / rel).read_bytes() == (b / rel).read_bytes(), rel
    assert pool.session_clock['loader_threads'] == loaders


@pytest.mark.gpu
def test_session_error_in_a_loader_thread_surfaces_and_does_not_hang(corpus, fixture, tmp_path):
    """A missing channel file: the reference dies in its load loop with the library's error;
    here the loader thread's exception reaches the caller (example named on stdout), the other
    threads are released and nothing is left in flight."""
    import copy
    enh = _enhancer(corpus, fixture, multiarray=True, wpe=False, bss_iterations=2)
    enh.loaders = 3
    examples = [copy.deepcopy(ex) for ex in list(enh.get_iterator('S02'))[:6]]
    array = sorted(examples[3]['audio_path']['observation'])[0]
    examples[3]['audio_path']['observation'][array][1] = str(tmp_path / 'missing.wav')
    (tmp_path / 'out' / 'dev').mkdir(parents=True)
    with pytest.raises(FileNotFoundError):
        enh._enhance_and_write(examples, tmp_path / 'out')
    assert len(list((tmp_path / 'out' / 'dev').glob('*.wav'))) <= 3


# ---------------------------------------------------------------- session_id=dev: S02 AND S09
# (scripts/run.py:45-71 resolves `dev` to both sessions; S09 has five arrays, mapping.py:67, so the
# channel count changes from 24 to 20 in the middle of a run.  Fixture chime5_dev_sessions.*:
# the reference's real session code on a two-session corpus, make_golden_session.py)
@pytest.fixture(scope='module')
def fixture_dev():
    return _load_fixture('chime5_dev_sessions')


@pytest.fixture(scope='module')
def corpus_dev(fixture_dev, tmp_path_factory):
    return _write_corpus(fixture_dev, tmp_path_factory.mktemp('chime5_dev_corpus'))


def test_dev_is_two_sessions_behind_one_database(corpus_dev, fixture_dev):
    from pb_chime5_amd.scripts.run import get_session_ids
    sessions = get_session_ids('dev')
    assert sessions == ['S02', 'S09'] == fixture_dev['corpus']['session_id']
    enh = _enhancer(corpus_dev, fixture_dev)
    it = enh.get_iterator(sessions)
    assert len(it) == len(fixture_dev['examples'])
    keys = ('start', 'end', 'num_samples', 'start_orig', 'end_orig', 'num_samples_orig')
    for ex, want in zip(it, fixture_dev['examples']):
        for key in ('example_id', 'speaker_id', 'session_id', 'reference_array') + keys:
            assert ex[key] == want[key], (ex['example_id'], key)
    arrays = {s: sorted(next(ex for ex in it if ex['session_id'] == s)['audio_path']['observation'])
              for s in sessions}
    assert len(arrays['S02']) == 6 and len(arrays['S09']) == 5 and 'U05' not in arrays['S09']
    # the activity of each session, both kept (loader threads of a two-session run ask for
    # either at any time)
    for session_id in sessions:
        activity = enh.activity[session_id]
        want = fixture_dev['activity'][session_id]
        assert list(activity.keys()) == list(want.keys())
        for array, tracks in want.items():
            for spk, intervals in tracks.items():
                got = [list(map(int, iv)) for iv in activity[array][spk].normalized_intervals]
                assert got == intervals, (session_id, array, spk)
    assert enh.activity['S02'] is enh.activity['S02'] and enh.activity['S09'] is enh.activity['S09']


@pytest.mark.gpu
def test_dev_examples_match_reference_at_24_and_20_channels(corpus_dev, fixture_dev):
    from tests.conftest import rel_err
    enh = _enhancer(corpus_dev, fixture_dev)
    it = enh.get_iterator(fixture_dev['corpus']['session_id'])
    gold = np.load(GOLDEN / 'chime5_dev_sessions.npz')
    channels = []
    for idx in fixture_dev['enhanced']:
        ex = it[idx]
        x_hat = enh.enhance_example(ex, debug=True)
        loc = enh.enhance_example_locals
        assert loc['obs'].shape == tuple(gold[f'obs_shape/{idx}'])
        channels.append(loc['obs'].shape[0])
        want = gold[f'x_hat/{idx}']
        assert x_hat.shape == want.shape
        assert rel_err(x_hat, want) < 1e-4
    assert channels == [24, 20]


@pytest.mark.gpu
@pytest.mark.parametrize('loaders,inflight', [(3, 2), (1, 3)])
def test_dev_session_run_switches_channel_count_mid_run(corpus_dev, fixture_dev, tmp_path,
                                                        loaders, inflight):
    """Enhancer.enhance_session(['S02', 'S09']) as `run.py with session_id=dev` calls it: the
    pipelined driver (loader threads, utterances in flight, staging blocks and device arenas
    re-sized when the observation goes from 24 to 20 channels and back -- longest-first order
    interleaves the sessions) writes every WAV byte-identical to the one-at-a-time loop."""
    sessions = fixture_dev['corpus']['session_id']
    a, b = tmp_path / 'seq', tmp_path / 'pipe'
    seq = _enhancer(corpus_dev, fixture_dev)
    seq.inflight = 1
    seq.enhance_session(sessions, a)
    pipe = _enhancer(corpus_dev, fixture_dev)
    pipe.inflight, pipe.loaders = inflight, loaders
    pipe.enhance_session(sessions, b)
    files = sorted(p.relative_to(a) for p in a.rglob('*.wav'))
    assert len(files) == len(fixture_dev['examples'])
    assert files == sorted(p.relative_to(b) for p in b.rglob('*.wav'))
    for rel in files:
        assert (a / rel).read_bytes() == (b / rel).read_bytes(), rel
    assert all(rel.parts[0] == 'dev' for rel in files)


@pytest.mark.gpu
def test_dev_session_run_with_two_ranks(corpus_dev, fixture_dev, tmp_path):
    """The same through the command line with two node-local ranks (longest first across BOTH
    sessions from the shared counter): every WAV once, byte-equal to one rank."""
    import os
    import subprocess
    import sys
    from conftest import REPO
    common = ['-m', 'pb_chime5_amd.scripts.run', 'with', f'database_path={corpus_dev}',
              'session_id=dev'] + [f'{k}={v}' for k, v in fixture_dev['enhancer'].items()]
    env = dict(os.environ, PYTHONPATH=str(REPO))
    one = subprocess.run([sys.executable] + common + ['-F', str(tmp_path / 'one')], cwd=str(REPO),
                         env=env, capture_output=True, text=True, timeout=900)
    assert one.returncode == 0, one.stderr[-3000:]
    two = subprocess.run([sys.executable, '-m', 'pb_chime5_amd.parallel', '-n', '2'] + common
                         + ['-F', str(tmp_path / 'two')], cwd=str(REPO), env=env,
                         capture_output=True, text=True, timeout=900)
    assert two.returncode == 0, two.stderr[-3000:]
    a, b = tmp_path / 'one' / '1' / 'audio', tmp_path / 'two' / '1' / 'audio'
    files = sorted(p.relative_to(a) for p in a.rglob('*.wav'))
    assert len(files) == len(fixture_dev['examples'])
    assert files == sorted(p.relative_to(b) for p in b.rglob('*.wav'))
    for rel in files:
        assert (a / rel).read_bytes() == (b / rel).read_bytes(), rel


# ---------------------------------------------------------------- command line
def test_cli_config_parsing():
    from pb_chime5_amd.scripts import run, kaldi_run, kaldi_run_rttm
    cfg = run.main(['print_config', 'with', 'session_id=S02', 'wpe=False', 'wpe_tabs=4',
                    'multiarray', 'bf=gev_ban', 'context_samples=16000'])
    assert cfg['session_id'] == 'S02' and cfg['wpe'] is False and cfg['wpe_tabs'] == 4
    assert cfg['multiarray'] is True and cfg['bf_drop_context'] is True and cfg['bf'] == 'gev_ban'
    assert cfg['bss_iterations'] == 20 and cfg['chime6'] is False       # reference defaults
    cfg = kaldi_run.main(['print_config', 'with', 'storage_dir=/x', 'job_id=3', 'number_of_jobs=8',
                          'multiarray=outer_array_mics'])
    assert (cfg['job_id'], cfg['number_of_jobs'], cfg['multiarray']) == (3, 8, 'outer_array_mics')
    cfg = kaldi_run_rttm.main(['print_config', 'with', 'storage_dir=/x', 'database_rttm=a.rttm'])
    assert cfg['activity_rttm'] == 'a.rttm' and cfg['multiarray'] == 'outer_array_mics'
    with pytest.raises(SystemExit):
        run.main(['print_config', 'with', 'no_such_key=1'])
    with pytest.raises(SystemExit):
        run.main(['frobnicate'])
    assert run.get_session_ids('dev') == ['S02', 'S09']
    assert run.get_session_ids(['eval', 'S03']) == ['S01', 'S03', 'S21']
    assert len(run.get_session_ids('all')) == 20


@pytest.mark.gpu
def test_cli_kaldi_run_static_split(corpus, fixture, tmp_path):
    """kaldi_run.py:59-90: job j of n writes examples j-1, j-1+n, ..."""
    from pb_chime5_amd.scripts import kaldi_run
    common = ['with', f'database_path={corpus}', 'session_id=S02', 'context_samples=8000',
              'multiarray=first_array_mics', 'wpe=False', 'bss_iterations=2',
              f'storage_dir={tmp_path}', 'number_of_jobs=3']
    kaldi_run.main(common + ['job_id=2'])
    ids = [ex['example_id'] for ex in fixture['examples']]
    written = sorted(p.stem for p in (tmp_path / 'audio' / 'dev').glob('*.wav'))
    assert written == sorted(ids[1::3])
    assert (tmp_path / 'sacred' / '1' / 'config.json').exists()
    with pytest.raises(AssertionError):
        kaldi_run.main(common + ['job_id=4'])


@pytest.mark.gpu
def test_cli_run_test_run(corpus, tmp_path):
    """run.py test_run: the first two examples into <file_storage>/<id>/audio."""
    from pb_chime5_amd.scripts import run
    run_dir = run.main(['test_run', 'with', f'database_path={corpus}', 'session_id=S02',
                        'context_samples=8000', 'wpe=False', 'bss_iterations=2',
                        'reference_array=U02', '-F', str(tmp_path / 'store')])
    assert run_dir == tmp_path / 'store' / '1'
    assert len(list((run_dir / 'audio' / 'dev').glob('*.wav'))) == 2


def test_session_tables_match_reference():
    """mapping.py builds the CHiME-5 session tables from compact rules; the expected dicts
    were dumped from the reference's pb_chime5/mapping.py:12-79 (tests/golden/mapping_tables.json)."""
    from pb_chime5_amd import mapping
    want = json.loads((GOLDEN / 'mapping_tables.json').read_text())
    assert dict(mapping.session_to_speakers) == want['session_to_speakers']
    assert dict(mapping.session_to_dataset) == want['session_to_dataset']
    assert dict(mapping.session_to_arrays) == want['session_to_arrays']
    with pytest.raises(KeyError):
        mapping.session_to_dataset['S99']


def test_activity_path_loads_pickles_written_by_the_reference(tmp_path):
    """Activity(type='path') (core.py:135-139) unpickles per-session dicts of the REFERENCE'S
    ArrayIntervall objects without that package.  tests/golden/activity_reference.pkl was
    written by the reference's own utils/intervall_array.py (make_golden_activity_pickle.py):
    its __reduce__ names the module-level `ArrayIntervall_from_str`; the JSON next
    to it holds what the reference's objects answer."""
    import pickle
    import shutil
    from pb_chime5_amd.core import Activity
    from pb_chime5_amd.utils.intervall_array import ArrayIntervall as Ours

    blob = (GOLDEN / 'activity_reference.pkl').read_bytes()
    assert b'pb_chime5.utils.intervall_array' in blob and b'ArrayIntervall_from_str' in blob
    assert b'pb_chime5_amd' not in blob
    shutil.copy(GOLDEN / 'activity_reference.pkl', tmp_path / 'S02.pkl')
    want = json.loads((GOLDEN / 'activity_reference.json').read_text())
    act = Activity(type='path', path=str(tmp_path))['S02']
    assert sorted(act) == sorted(want)
    for array, speakers in want.items():
        assert list(act[array]) == list(speakers)          # dict order = class order
        for speaker, w in speakers.items():
            got = act[array][speaker]
            assert isinstance(got, Ours)
            assert list(got.shape) == w['shape']
            assert [list(i) for i in got.normalized_intervals] == w['intervals']
            for sl in w['slices']:
                dense = got[sl['start']:sl['stop']]
                assert dense.dtype == bool and dense.shape == (sl['stop'] - sl['start'],)
                edges = np.diff(np.concatenate([[0], dense.astype(np.int8), [0]]))
                runs = [list(map(int, r)) for r in zip(np.flatnonzero(edges > 0),
                                                       np.flatnonzero(edges < 0))]
                assert runs == sl['runs'], (array, speaker, sl['start'])
    # and what this package pickles loads again (same protocol both ways)
    again = pickle.loads(pickle.dumps(act['U01']['P05']))
    assert again.normalized_intervals == act['U01']['P05'].normalized_intervals


def test_activity_cache_hands_every_thread_the_session_it_asked_for():
    """`Activity.__getitem__` is called from every loader thread of a session; a rank of a
    multi-session run sees its sessions interleaved (the one-entry cache could
    return another thread's session)."""
    import threading
    import time
    from pb_chime5_amd.core import Activity

    computed = []

    class Slow(Activity):
        def _annotation_activity(self, session_id):
            computed.append(session_id)
            time.sleep(0.002)
            return {'session': session_id}

    act = Slow()
    errors = []

    def work(seed):
        rng = np.random.default_rng(seed)
        try:
            for _ in range(150):
                sid = f'S{int(rng.integers(0, 3)):02d}'
                assert act[sid]['session'] == sid
        except Exception as e:       # noqa: BLE001
            errors.append(e)
    threads = [threading.Thread(target=work, args=(s,)) for s in range(6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert sorted(computed) == ['S00', 'S01', 'S02']          # once each, not once per switch
    for sid in ('S03', 'S04', 'S05', 'S06'):                   # bounded
        act[sid]
    assert len(act._cache) == Activity._CACHE_SESSIONS and 'S00' not in act._cache
