"""Generate the golden fixtures in this directory.

Runs ONLY in the build container, where the reference checkout is mounted at
/root/reference.  It copies the reference package to a scratch directory,
cythonises its one hot-path-adjacent extension, registers stub modules for the
third-party packages that are not installable here, and
then drives the reference's REAL orchestration code
(``pb_chime5/core.py``, ``speech_enhancement/beamforming_wrapper.py``,
``database/chime5/database.py``, ``utils/numpy_utils.py``,
``utils/intervall_array.py``, ``math/solve.py``) on small seeded inputs.

The numeric third-party calls (nara_wpe / pb_bss) are delegated to the CPU oracle
(``oracle/gss_oracle.py``), so the captured vectors pin everything the
reference itself owns -- activity/indexing logic, layouts, mask post-processing,
dispatch, context bookkeeping -- bit-exactly, and give end-to-end expected
outputs "reference glue + oracle arithmetic".

Nothing from /root/reference is written into the repository: the fixtures are
inputs and outputs only.

Usage:  python tests/golden/make_golden.py
"""
import json
import shutil
import subprocess
import sys
import tempfile
import types
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
REFERENCE = Path('/root/reference')

sys.path.insert(0, str(REPO / 'oracle'))
sys.path.insert(0, str(REPO))
import gss_oracle as oracle  # noqa: E402


def _prepare_reference(scratch):
    ref = scratch / 'ref'
    shutil.copytree(REFERENCE / 'pb_chime5', ref / 'pb_chime5')
    utils = ref / 'pb_chime5' / 'utils'
    ext = scratch / 'ext'
    ext.mkdir()
    shutil.copy(utils / 'intervall_array_util.pyx', ext)
    setup = ext / 'setup_ext.py'
    setup.write_text(
        "from setuptools import setup\n"
        "from Cython.Build import cythonize\n"
        "setup(ext_modules=cythonize(['intervall_array_util.pyx'],"
        " language_level=3))\n")
    subprocess.run([sys.executable, str(setup), 'build_ext', '--inplace'],
                   cwd=ext, check=True, capture_output=True)
    for so in ext.glob('intervall_array_util*.so'):
        shutil.copy(so, utils)
    return ref


def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _cached_property:
    def __init__(self, func):
        self.func = func
        self.name = func.__name__

    def __get__(self, obj, cls):
        if obj is None:
            return self
        value = obj.__dict__[self.name] = self.func(obj)
        return value


class _Trainer:
    """Stands in for pb_bss.distribution.CACGMMTrainer; delegates to the oracle."""

    def fit(self, y, initialization=None, iterations=100,
            source_activity_mask=None, **kw):
        assert not kw, kw
        return oracle.CACGMMTrainer().fit(
            y, initialization, iterations=iterations,
            source_activity_mask=source_activity_mask)


def _register_stubs():
    np.int = int
    np.object = object
    np.float = float
    _module('cached_property', cached_property=_cached_property)
    _module('dlp_mpi', IS_MASTER=True, MASTER=0, barrier=lambda: None,
            bcast=lambda x, *a, **k: x,
            split_managed=lambda it, **k: iter(it))
    _module('dlp_mpi.util', ensure_single_thread_numeric=lambda: None)
    nara = _module('nara_wpe')
    nara.wpe = _module('nara_wpe.wpe', wpe_v8=oracle.wpe_v8)
    nara.utils = _module(
        'nara_wpe.utils', stft=oracle.stft, istft=oracle.istft,
        _samples_to_stft_frames=oracle.samples_to_stft_frames)
    pb = _module('pb_bss')
    pb.distribution = _module('pb_bss.distribution', CACGMMTrainer=_Trainer)
    _module('pb_bss.distribution.utils', stack_parameters=lambda x: x)
    bf = _module(
        'pb_bss.extraction.beamformer',
        get_power_spectral_density_matrix=oracle.get_power_spectral_density_matrix,
        get_mvdr_vector_souden=oracle.get_mvdr_vector_souden,
        blind_analytic_normalization=oracle.blind_analytic_normalization,
        apply_beamforming_vector=oracle.apply_beamforming_vector,
        get_gev_vector=None, get_lcmv_vector_souden=None)
    pb.extraction = _module('pb_bss.extraction', beamformer=bf)
    _module('pb_bss.extraction.mask_module', lorenz_mask=None,
            quantile_mask=None)
    _module('lazy_dataset', from_dict=None, concatenate=None)
    _module('lazy_dataset.database', Database=object)
    _module('soundfile')


def _save(name, **arrays):
    np.savez_compressed(HERE / name, **arrays)
    size = (HERE / name).stat().st_size
    print(f'{name}: {size / 1024:.0f} KiB')


def _small_inputs(seed, D, N, K, context):
    """Seeded, self-contained inputs (stored in the fixture, so the generator of
    pb_chime5_amd.synthetic is not part of the contract)."""
    rng = np.random.default_rng(seed)
    src = rng.standard_normal((K - 1, N))
    mix = rng.standard_normal((D, K - 1)) + 0.5
    act = np.zeros((K, N), dtype=bool)
    obs = 0.05 * rng.standard_normal((D, N))
    for k in range(K - 1):
        a = int(rng.integers(0, N // 3))
        b = int(rng.integers(2 * N // 3, N))
        act[k, a:b] = True
        h = rng.standard_normal((D, 40)) * np.exp(-np.arange(40) / 8.0)
        for d in range(D):
            obs[d] += mix[d, k] * np.convolve(src[k] * act[k], h[d])[:N]
    act[0, context:N - context] = True
    act[-1] = True
    ex = {'start': {'original': 1000}, 'start_orig': {'original': 1000 + context},
          'end_orig': {'original': 1000 + N - context},
          'end': {'original': 1000 + N}}
    return obs, act, ex


def make_orchestration(core):
    """Reference Enhancer.enhance_observation driven on small inputs."""
    cases = {}

    def run(tag, obs, act, ex, speaker, enhancer_kwargs, stft=(64, 16)):
        names = [f'P{k + 1:02d}' for k in range(act.shape[0] - 1)] + ['Noise']
        activity = dict(zip(names, act))
        enh = core.get_enhancer(stft_size=stft[0], stft_shift=stft[1],
                                **enhancer_kwargs)
        x_hat = enh.enhance_observation(obs, ex_array_activity=activity,
                                        speaker_id=speaker, ex=ex, debug=True)
        loc = enh.enhance_observation_locals
        out = {
            'obs': obs, 'activity': act, 'x_hat': x_hat,
            'acitivity_freq': loc['acitivity_freq'],
            'masks': loc['masks'], 'target_mask': loc['target_mask'],
            'distortion_mask': loc['distortion_mask'],
            'target_speaker_index': np.int64(loc['target_speaker_index']),
            'X_hat': loc['X_hat'],
            'Obs': loc['Obs'],
            'ex': np.array([ex['start']['original'],
                            ex['start_orig']['original'],
                            ex['end_orig']['original'],
                            ex['end']['original']]),
            'stft': np.array(stft),
        }
        if enhancer_kwargs.get('bf_drop_context', True):
            out['context_frames'] = np.array(
                [loc['start_context_frames'], loc['end_context_frames']])
        gl = getattr(enh.gss_block, 'locals', None)
        if gl is not None:
            out['gss_initialization'] = gl['initialization'][0]
            out['gss_source_active_mask'] = gl['source_active_mask'][0]
        if tag not in ('default', 'nowpe', 'ragged', 'stft1024'):
            # same inputs as 'default': keep the outputs only
            for k in ('obs', 'activity', 'Obs', 'acitivity_freq',
                      'gss_initialization', 'gss_source_active_mask'):
                out.pop(k, None)
        for k, v in out.items():
            cases[f'{tag}/{k}'] = np.asarray(v)
        cases[f'{tag}/kwargs'] = np.array(json.dumps(enhancer_kwargs))
        cases[f'{tag}/speaker'] = np.array(speaker)

    obs, act, ex = _small_inputs(11, D=4, N=1200, K=3, context=200)
    base = dict(wpe=True, wpe_tabs=2, wpe_delay=2, wpe_iterations=2,
                bss_iterations=3, context_samples=200)
    run('default', obs, act, ex, 'P01', base)
    run('target2', obs, act, ex, 'P02', base)
    run('nowpe', obs, act, ex, 'P01', dict(base, wpe=False))
    run('nodrop', obs, act, ex, 'P01', dict(base, bf_drop_context=False))
    run('maskmul', obs, act, ex, 'P01', dict(base, postfilter='mask_mul'))
    run('ch2', obs, act, ex, 'P01', dict(base, bf='ch2'))
    run('sum', obs, act, ex, 'P01', dict(base, bf='sum'))
    run('post0', obs, act, ex, 'P01', dict(base, bss_iterations_post=0))
    run('post2', obs, act, ex, 'P01', dict(base, bss_iterations_post=2))
    # no context at all: the reference still zeroes 3 frames on either side
    ex0 = {'start': {'original': 0}, 'start_orig': {'original': 0},
           'end_orig': {'original': 1200}, 'end': {'original': 1200}}
    run('ctx0', obs, act, ex0, 'P01', dict(base, context_samples=0))
    # N not a multiple of the shift, 5 channels, 4 classes, short burst
    obs2, act2, ex2 = _small_inputs(12, D=5, N=1131, K=4, context=100)
    act2[1] = False
    act2[1, 500:505] = True            # burst shorter than a window
    run('ragged', obs2, act2, ex2, 'P03', dict(base, bss_iterations=2))
    _save('orchestration_small.npz', **cases)

    # reference default STFT geometry (1024/256, F = 513)
    cases = {}
    obs3, act3, ex3 = _small_inputs(13, D=3, N=5000, K=3, context=1000)
    run('stft1024', obs3, act3, ex3, 'P01',
        dict(wpe=True, wpe_tabs=2, wpe_iterations=1, bss_iterations=2,
             context_samples=1000), stft=(1024, 256))
    keep = {}
    for k, v in cases.items():
        if v.ndim >= 2 and v.shape[-1] == 513:
            v = v[..., ::16]            # every 16th bin keeps the file small
        keep[k] = v
    _save('orchestration_1024.npz', **keep)


def make_context_frames(core):
    rows = []
    for size, shift in ((1024, 256), (512, 128), (64, 16)):
        for fading in (True, False):
            for s in (0, 1, 255, 256, 257, 1000, 16000, 160000, 240000):
                for e in (0, 300, 240000):
                    if not fading and (s < size - shift or e < size - shift):
                        continue
                    ex = {'start': {'original': 5}, 'start_orig': {'original': 5 + s},
                          'end_orig': {'original': 100000 + s},
                          'end': {'original': 100000 + s + e}}
                    a, b = core.start_end_context_frames(ex, size, shift, fading)
                    rows.append([size, shift, int(fading), s, e, a, b])
    return np.array(rows, dtype=np.int64)


def make_beamformer(bw):
    rng = np.random.default_rng(21)
    D, T, F = 4, 40, 9
    Y = rng.standard_normal((D, T, F)) + 1j * rng.standard_normal((D, T, F))
    X3 = rng.uniform(size=(D, T, F))
    N3 = 1 - X3
    out = {'Y': Y, 'X_mask3': X3, 'N_mask3': N3}
    X2, N2 = np.median(X3, axis=0), np.median(N3, axis=0)
    out['ban_2d'] = bw.beamform_mvdr_souden_from_masks(Y, X2, N2, ban=True)
    out['noban_2d'] = bw.beamform_mvdr_souden_from_masks(Y, X2, N2, ban=False)
    out['ban_3d'] = bw.beamform_mvdr_souden_from_masks(Y, X3, N3, ban=True)
    out['ban_4d'] = bw.beamform_mvdr_souden_from_masks(
        Y[None], X3[None], N3[None], ban=True)
    bf = bw._Beamformer(Y, X2, N2)
    out['Y_FDT'] = bf.Y
    out['X_mask_FT'] = bf.X_mask
    out['Cov_X'] = bf._Cov_X
    out['Cov_N'] = bf._Cov_N
    out['w_souden'] = bf._w_mvdr_souden
    out['w_souden_ban'] = bf._w_mvdr_souden_ban
    # a bin whose distortion mask is all zero -> singular Cov_N -> lstsq path
    N2z = N2.copy()
    N2z[:, 3] = 0
    bfz = bw._Beamformer(Y, X2, N2z)
    out['N_mask_zero_bin'] = N2z
    out['w_souden_zero_bin'] = bfz._w_mvdr_souden
    return out


def make_wpe_block(core):
    rng = np.random.default_rng(31)
    A, C, T, F = 2, 2, 30, 5
    Obs = rng.standard_normal((A, C, T, F)) + 1j * rng.standard_normal((A, C, T, F))
    w = core.WPE(taps=2, delay=1, iterations=2, psd_context=0)
    return {
        'Obs': Obs,
        'stack_true': w(Obs, stack=True),
        'stack_false': w(Obs, stack=False),
        'ndim3': w(Obs[0]),
    }


def make_host_helpers(ref_pkg):
    from pb_chime5.database.chime5.database import (
        activity_time_to_frequency, activity_frequency_to_time,
        _adjust_start_end)
    from pb_chime5.utils.numpy_utils import segment_axis_v2, pad_axis, morph
    from pb_chime5.utils.intervall_array import ArrayIntervall
    from pb_chime5.math.solve import stable_solve
    out = {}
    rng = np.random.default_rng(41)

    # activity_time_to_frequency, edge cases
    cases = []
    for i, (n, size, shift, fading, pad) in enumerate([
            (14, 4, 2, True, True), (14, 4, 2, False, True),
            (2000, 64, 16, True, True), (2001, 64, 16, True, True),
            (5000, 1024, 256, True, True), (5120, 1024, 256, True, True),
            (80000, 1024, 256, True, True), (3000, 1024, 256, False, False),
            (500, 1024, 256, True, True)]):
        act = rng.uniform(size=(3, n)) < 0.002
        act[1] = False
        act[1, :1] = True
        act[1, -1:] = True
        act[2, n // 2:n // 2 + 3] = True
        res = activity_time_to_frequency(act, size, shift, fading, stft_pad=pad)
        out[f'a2f/{i}/act'] = act
        out[f'a2f/{i}/par'] = np.array([size, shift, int(fading), int(pad)])
        out[f'a2f/{i}/res'] = res
    vad = np.array([0, 1, 0, 1, 0, 0, 1, 0, 0])
    out['f2a/vad'] = vad
    out['f2a/res'] = activity_frequency_to_time(vad, 4, 2, False)

    # segment_axis_v2 / pad_axis
    for i, (n, length, shift, end) in enumerate([
            (10, 4, 2, 'pad'), (5, 4, 2, 'pad'), (5, 4, 2, 'cut'),
            (6, 4, 2, 'pad'), (7, 8, 1, 'pad'), (9, 8, 1, 'pad'),
            (9, 8, 2, 'cut'), (7, 8, 2, 'cut'), (100, 16, 4, 'pad'),
            (101, 16, 4, 'pad'), (3, 16, 4, 'pad')]):
        x = np.arange(2 * n).reshape(2, n)
        out[f'seg/{i}/par'] = np.array([n, length, shift, end == 'pad'])
        out[f'seg/{i}/res'] = np.array(segment_axis_v2(x, length, shift, end=end))
    out['pad_axis/a'] = pad_axis(np.ones([3, 4]), (1, 2), axis=1)
    out['pad_axis/b'] = pad_axis(np.ones([3, 4]), 1, axis=0)

    # morph layouts used on the hot path
    x = rng.standard_normal((3, 4, 5))
    out['morph/x'] = x
    out['morph/DTF->FDT'] = morph('DTF->FDT', x)
    out['morph/ACN->A*CN'] = morph('ACN->A*CN', x)
    out['morph/A*CTF->ACTF'] = morph('A*CN->ACN', morph('ACN->A*CN', x), A=3)
    out['morph/TF->FT'] = morph('TF->FT', x[0], reduce=np.median)
    out['morph/DTF->FT'] = morph('DTF->FT', x, reduce=np.median)
    out['morph/1DTF->FDT'] = morph('1DTF->FDT', x[None])

    # ArrayIntervall
    ai = ArrayIntervall(50)
    ai[10:20] = 1
    ai[25:30] = 1
    out['ai/0'] = ai[19:26]
    ai[5:10] = 1
    ai[10:13] = np.array([False, True, False])
    out['ai/repr'] = np.array(repr(ai))
    out['ai/1'] = ai[:]
    out['ai/2'] = ai[3:40]
    out['ai/normalized'] = np.array(ai.normalized_intervals)
    out['ai/from_array'] = np.array(ArrayIntervall.from_array(
        np.array([1, 1, 0, 1, 0, 0, 1, 1, 0], dtype=bool)).normalized_intervals)

    # _adjust_start_end
    rows = []
    for ws, we, as_, ae in [(10, 20, 10, 19), (10, 20, 10, 21), (5, 50, 7, 52),
                            (5, 50, 7, 55), (5, 50, 7, 47), (0, 9, 100, 104)]:
        rows.append([ws, we, as_, ae, *_adjust_start_end(ws, we, as_, ae)])
    out['adjust_start_end'] = np.array(rows)

    # stable_solve: regular, one singular, all-zero
    np.linalg.linalg = np.linalg      # alias removed in NumPy 2
    A = rng.standard_normal((3, 6, 6)) + 1j * rng.standard_normal((3, 6, 6))
    B = rng.standard_normal((3, 6, 6)) + 1j * rng.standard_normal((3, 6, 6))
    out['solve/A'] = A
    out['solve/B'] = B
    out['solve/regular'] = stable_solve(A, B)
    A2 = A.copy()
    A2[2, 3, :] = 0
    out['solve/A_singular'] = A2
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        out['solve/singular'] = stable_solve(A2, B)
        out['solve/zero'] = stable_solve(np.zeros((6, 6), complex),
                                         np.zeros((6, 6), complex))
    return out


def make_enhance_example(core):
    """Reference Enhancer.enhance_example with in-memory audio and activity."""
    from pb_chime5.utils.intervall_array import ArrayIntervall
    rng = np.random.default_rng(51)
    arrays = ['U01', 'U02', 'U03']
    total = 3000
    audio = {a: 0.1 * rng.standard_normal((4, total + i * 7))
             for i, a in enumerate(arrays)}

    def load_audio(path, start=None, stop=None):
        return audio[path][:, start:stop]
    core.load_audio = load_audio

    speakers = ['P05', 'P06', 'Noise']
    act = {}
    for a in arrays:
        act[a] = {}
        for i, s in enumerate(speakers):
            ai = ArrayIntervall(total + 500)
            if s == 'Noise':
                ai[0:total + 500] = 1
            else:
                ai[700 + 300 * i:1600 + 400 * i] = 1
                ai[2300:2300 + 100 * (i + 1)] = 1
            act[a][s] = ai
    activity = {'S99': act}

    out = {}
    for a in arrays:
        out[f'audio/{a}'] = audio[a]
    out['speakers'] = np.array(speakers)
    out['intervals/P05'] = np.array(act['U01']['P05'].normalized_intervals)
    out['intervals/P06'] = np.array(act['U01']['P06'].normalized_intervals)
    ctx = 400
    ex = {
        'session_id': 'S99', 'speaker_id': 'P06', 'example_id': 'P06_S99_x',
        'reference_array': 'U02',
        'audio_path': {'observation': {a: a for a in arrays}},
        'start': {'original': 1000 - ctx,
                  'observation': {'U01': 1000 - ctx, 'U02': 1003 - ctx,
                                  'U03': 998 - ctx}},
        'end': {'original': 2200 + ctx,
                'observation': {'U01': 2200 + ctx, 'U02': 2203 + ctx,
                                'U03': 2190 + ctx}},
        'start_orig': {'original': 1000,
                       'observation': {'U01': 1000, 'U02': 1003, 'U03': 998}},
        'end_orig': {'original': 2200,
                     'observation': {'U01': 2200, 'U02': 2203, 'U03': 2190}},
        'num_samples_orig': {'original': 1200,
                             'observation': {'U01': 1200, 'U02': 1200,
                                             'U03': 1192}},
    }
    out['ex'] = np.array(json.dumps(ex))
    for tag, multiarray in [('true', True), ('outer', 'outer_array_mics'),
                            ('first', 'first_array_mics'), ('false', False)]:
        enh = core.get_enhancer(multiarray=multiarray, context_samples=ctx,
                                wpe=True, wpe_tabs=2, wpe_iterations=1,
                                bss_iterations=2, stft_size=64, stft_shift=16)
        enh.activity = activity
        x_hat = enh.enhance_example(ex, debug=True)
        loc = enh.enhance_example_locals
        out[f'{tag}/x_hat'] = x_hat
        out[f'{tag}/obs'] = loc['obs']
        out[f'{tag}/activity'] = np.array(list(loc['ex_array_activity'].values()))
    return out


def _chime5_example(shift=0):
    arrays = ['U01', 'U02', 'U04']
    ex = {
        'audio_path': {'observation': {a: [f'{a}.CH{c}.wav' for c in range(1, 5)] for a in arrays},
                       'worn_microphone': {'P05': 'P05.wav', 'P06': 'P06.wav'}},
        'start': {'original': 649600 + shift,
                  'observation': {'U01': 650080 + shift, 'U02': 650123 + shift, 'U04': 649001 + shift},
                  'worn_microphone': {'P05': 649600 + shift, 'P06': 649700 + shift}},
        'end': {'original': 701120 + shift,
                'observation': {'U01': 701600 + shift, 'U02': 701650 + shift, 'U04': 700519 + shift},
                'worn_microphone': {'P05': 701120 + shift, 'P06': 701110 + shift}},
    }
    ex['num_samples'] = {
        k: (ex['end'][k] - ex['start'][k] if k == 'original' else
            {a: ex['end'][k][a] - ex['start'][k][a] for a in ex['start'][k]})
        for k in ex['start']}
    return ex


def make_context_bookkeeping():
    """Reference backup_orig_start_end / adjust_start_end / AddContext on CHiME-5 shaped
    examples (database.py:540-570, 706-1053)."""
    import copy
    from pb_chime5.database.chime5.database import (
        backup_orig_start_end, adjust_start_end, AddContext)
    cases = {}
    for name, shift, samples, equal, adjust in [
            ('plain', 0, 240000, False, False), ('equal', 0, 240000, True, False),
            ('adjust_equal', 0, 240000, True, True), ('early', -640000, 16000, True, True),
            ('early_plain', -640000, 16000, False, False), ('tuple', 0, (1000, 3), False, True)]:
        ex = _chime5_example(shift)
        inp = copy.deepcopy(ex)
        ex = backup_orig_start_end(ex)
        if adjust:
            ex = adjust_start_end(ex)
        ex = AddContext(samples, equal_start_context=equal)(ex)
        cases[name] = {'input': inp, 'samples': samples, 'equal': equal, 'adjust': adjust,
                       'output': {k: ex[k] for k in ('start', 'end', 'num_samples', 'start_orig',
                                                     'end_orig', 'num_samples_orig')}}
    (HERE / 'context_bookkeeping.json').write_text(json.dumps(cases, indent=1, default=int))
    print('context_bookkeeping.json')


def main():
    with tempfile.TemporaryDirectory() as tmp:
        ref = _prepare_reference(Path(tmp))
        _register_stubs()
        sys.path.insert(0, str(ref))
        # pb_chime5/__init__.py pulls in activity.py / database JSON handling
        # that is not on the hot path; import the needed modules directly.
        pkg = types.ModuleType('pb_chime5')
        pkg.__path__ = [str(ref / 'pb_chime5')]
        pkg.git_root = ref
        sys.modules['pb_chime5'] = pkg
        import pb_chime5.core as core
        import pb_chime5.speech_enhancement.beamforming_wrapper as bw

        make_orchestration(core)
        _save('context_frames.npz', rows=make_context_frames(core))
        _save('beamformer.npz', **make_beamformer(bw))
        _save('wpe_block.npz', **make_wpe_block(core))
        _save('host_helpers.npz', **make_host_helpers(ref))
        _save('enhance_example.npz', **make_enhance_example(core))
        make_context_bookkeeping()


if __name__ == '__main__':
    main()
