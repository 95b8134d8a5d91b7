"""Pins the CPU oracle (and the product's host-side helpers) against everything the
reference itself can vouch for:

* known-answer vectors of the reference's doctests
  (database/chime5/database.py:417-453, utils/numpy_utils.py:42-136,500-548,
  utils/intervall_array.py, math/solve.py:38-87), and
* fixtures captured by running the reference's own orchestration code
  (core.py, beamforming_wrapper.py) in the build container
  (tests/golden/make_golden.py).

No GPU needed.
"""
import json

import numpy as np
import pytest

import gss_oracle as oracle
from conftest import rel_err


# ---------------------------------------------------------------- reference doctest vectors
SIGNAL = np.array([0, 0, 0, 0, 0, 1, -3, 0, 5, 0, 0, 0, 0, 0], dtype=float)
VAD = np.array([0, 0, 0, 0, 0, 1, 1, 0, 1, 0, 0, 0, 0, 0])
STFT_FADING = np.array([
    [0, 0, 0], [0, 0, 0], [1, 1j, -1], [-2, 3 - 1j, -4], [2, -8, 2], [5, 5, 5], [0, 0, 0],
    [0, 0, 0]], dtype=complex)


def test_stft_known_answers_from_reference_doctest():
    got = oracle.stft(SIGNAL, size=4, shift=2, fading=True, window=np.ones(4))
    assert np.allclose(got, STFT_FADING, atol=1e-15)
    got = oracle.stft(SIGNAL, size=4, shift=2, fading=False, window=np.ones(4))
    assert np.allclose(got, STFT_FADING[1:-1], atol=1e-15)
    assert oracle.stft(np.zeros(200000), size=1024, shift=256, fading=False,
                       pad=False).shape == (778, 513)


@pytest.mark.parametrize('impl', ['oracle', 'product'])
def test_activity_time_to_frequency_known_answers(impl):
    if impl == 'oracle':
        fn = oracle.activity_time_to_frequency
    else:
        from pb_chime5_amd.database.chime5 import activity_time_to_frequency as fn
    want = np.array([False, False, True, True, True, True, False, False])
    assert np.array_equal(fn(VAD, 4, 2, True), want)
    assert np.array_equal(fn([VAD, VAD], 4, 2, True), np.array([want, want]))
    assert np.array_equal(fn(VAD, 4, 2, False), want[1:-1])
    assert fn(np.zeros(200000), 1024, 256, False, stft_pad=False).shape == (778,)


@pytest.mark.parametrize('impl', ['oracle', 'product'])
def test_activity_fixture_bit_exact(golden, impl):
    if impl == 'oracle':
        fn = oracle.activity_time_to_frequency
    else:
        from pb_chime5_amd.database.chime5 import activity_time_to_frequency as fn
    g = golden('host_helpers.npz')
    i = 0
    while f'a2f/{i}/res' in g.files:
        size, shift, fading, pad = (int(v) for v in g[f'a2f/{i}/par'])
        got = fn(g[f'a2f/{i}/act'], size, shift, bool(fading), stft_pad=bool(pad))
        assert got.dtype == bool and np.array_equal(got, g[f'a2f/{i}/res']), i
        i += 1
    assert i >= 9


def test_activity_frequency_to_time_fixture(golden):
    from pb_chime5_amd.database.chime5 import activity_frequency_to_time
    g = golden('host_helpers.npz')
    assert np.array_equal(activity_frequency_to_time(g['f2a/vad'], 4, 2, False), g['f2a/res'])


def test_segment_axis_fixture(golden):
    from pb_chime5_amd.utils.numpy_utils import segment_axis_v2
    g = golden('host_helpers.npz')
    i = 0
    while f'seg/{i}/res' in g.files:
        n, length, shift, pad = (int(v) for v in g[f'seg/{i}/par'])
        x = np.arange(2 * n).reshape(2, n)
        end = 'pad' if pad else 'cut'
        want = g[f'seg/{i}/res']
        assert np.array_equal(segment_axis_v2(x, length, shift, end=end), want), i
        assert np.array_equal(oracle.segment_axis(x, length, shift, end=end), want), i
        i += 1
    assert i >= 10
    # reference doctests (numpy_utils.py:42-60)
    assert np.array_equal(segment_axis_v2(np.arange(10), 4, 2),
                          [[0, 1, 2, 3], [2, 3, 4, 5], [4, 5, 6, 7], [6, 7, 8, 9]])
    assert np.array_equal(segment_axis_v2(np.arange(10), 4, -2),
                          [[6, 7, 8, 9], [4, 5, 6, 7], [2, 3, 4, 5], [0, 1, 2, 3]])
    assert np.array_equal(segment_axis_v2(np.arange(5), 4, 2, axis=0, end='pad'),
                          [[0, 1, 2, 3], [2, 3, 4, 0]])
    assert segment_axis_v2(np.arange(7), 8, 2, axis=0, end='cut').shape == (0, 8)
    assert np.array_equal(segment_axis_v2(np.arange(5), 3, 1, end='conv_pad')[0], [0, 0, 0])


def test_pad_axis_and_morph_fixture(golden):
    from pb_chime5_amd.utils.numpy_utils import pad_axis, morph
    g = golden('host_helpers.npz')
    assert np.array_equal(pad_axis(np.ones([3, 4]), (1, 2), axis=1), g['pad_axis/a'])
    assert np.array_equal(pad_axis(np.ones([3, 4]), 1, axis=0), g['pad_axis/b'])
    x = g['morph/x']
    assert np.array_equal(morph('DTF->FDT', x), g['morph/DTF->FDT'])
    assert np.array_equal(morph('ACN->A*CN', x), g['morph/ACN->A*CN'])
    assert np.array_equal(morph('A*CN->ACN', morph('ACN->A*CN', x), A=3), g['morph/A*CTF->ACTF'])
    assert np.array_equal(morph('TF->FT', x[0], reduce=np.median), g['morph/TF->FT'])
    assert np.array_equal(morph('DTF->FT', x, reduce=np.median), g['morph/DTF->FT'])
    assert np.array_equal(morph('1DTF->FDT', x[None]), g['morph/1DTF->FDT'])


def test_array_intervall_fixture(golden):
    from pb_chime5_amd.utils.intervall_array import ArrayIntervall
    g = golden('host_helpers.npz')
    ai = ArrayIntervall(50)
    ai[10:20] = 1
    ai[25:30] = 1
    assert np.array_equal(ai[19:26], g['ai/0'])
    ai[5:10] = 1
    ai[10:13] = np.array([False, True, False])
    assert repr(ai) == str(g['ai/repr'])
    assert np.array_equal(ai[:], g['ai/1'])
    assert np.array_equal(ai[3:40], g['ai/2'])
    assert np.array_equal(np.array(ai.normalized_intervals), g['ai/normalized'])
    a = np.array([1, 1, 0, 1, 0, 0, 1, 1, 0], dtype=bool)
    assert np.array_equal(np.array(ArrayIntervall.from_array(a).normalized_intervals),
                          g['ai/from_array'])
    assert np.array_equal(ArrayIntervall.from_array(a)[:], a)
    # reference doctests (intervall_array.py:14-21, 302-332)
    assert repr(ArrayIntervall.from_str('1:4, 5:20, 21:25', shape=50)) == \
        'ArrayIntervall("1:4, 5:20, 21:25", shape=(50,))'
    b = ArrayIntervall(50)
    for s, e in ((10, 15), (5, 10), (1, 4), (15, 20), (21, 25), (10, 15)):
        b[s:e] = 1
    assert repr(b) == 'ArrayIntervall("1:4, 5:20, 21:25", shape=(50,))'
    b[0:50] = 1
    b[3:6] = np.array([True, False, True])
    assert repr(b) == 'ArrayIntervall("0:4, 5:50", shape=(50,))'
    b[10:13] = np.array([False, True, False])
    assert repr(b) == 'ArrayIntervall("0:4, 5:10, 11:12, 13:50", shape=(50,))'
    assert len(b) == 50
    import pickle
    assert repr(pickle.loads(pickle.dumps(b))) == repr(b)


def test_adjust_start_end_fixture(golden):
    from pb_chime5_amd.database.chime5 import _adjust_start_end
    for ws, we, a_s, a_e, ns, ne in golden('host_helpers.npz')['adjust_start_end']:
        assert _adjust_start_end(ws, we, a_s, a_e) == (ns, ne)


def test_stable_solve_fixture(golden):
    g = golden('host_helpers.npz')
    assert rel_err(oracle.stable_solve(g['solve/A'], g['solve/B']), g['solve/regular']) < 1e-12
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        assert rel_err(oracle.stable_solve(g['solve/A_singular'], g['solve/B']),
                       g['solve/singular']) < 1e-10
    z = oracle.stable_solve(np.zeros((6, 6), complex), np.zeros((6, 6), complex))
    assert np.array_equal(z, g['solve/zero']) and not z.any()


def test_context_frames_fixture(golden):
    from pb_chime5_amd import core
    rows = golden('context_frames.npz')['rows']
    assert len(rows) > 100
    for size, shift, fading, s, e, a, b in rows:
        ex = {'start': {'original': 5}, 'start_orig': {'original': 5 + s},
              'end_orig': {'original': 100000 + s}, 'end': {'original': 100000 + s + e}}
        assert oracle.start_end_context_frames(ex, size, shift, bool(fading)) == (a, b)
        # product: same integers (through the C ABI helper, no GPU needed)
        assert core.start_end_context_frames(ex, int(size), int(shift), bool(fading)) == (a, b)
    assert oracle.samples_to_stft_frames(16000, 1024, 256, fading=True) == 66
    assert oracle.samples_to_stft_frames(240000, 1024, 256, fading=True) == 941


# ---------------------------------------------------------------- reference orchestration
def _ex(vec):
    return {'start': {'original': int(vec[0])}, 'start_orig': {'original': int(vec[1])},
            'end_orig': {'original': int(vec[2])}, 'end': {'original': int(vec[3])}}


@pytest.mark.parametrize('fixture', ['orchestration_small.npz', 'orchestration_1024.npz'])
def test_oracle_glue_equals_reference_glue(golden, fixture):
    """oracle.enhance_observation (restated glue) against the reference's real
    Enhancer.enhance_observation driving the same arithmetic: identical bits for
    everything integer / bool, identical floats up to summation order."""
    g = golden(fixture)
    tags = sorted({k.split('/')[0] for k in g.files})
    assert len(tags) >= 1
    for tag in tags:
        kw = json.loads(str(g[f'{tag}/kwargs']))
        src = tag if f'{tag}/obs' in g.files else 'default'
        obs, act = g[f'{src}/obs'], g[f'{src}/activity']
        size, shift = (int(v) for v in g[f'{tag}/stft'])
        target = int(g[f'{tag}/target_speaker_index'])
        x_hat, det = oracle.enhance_observation(
            obs, act, target, _ex(g[f'{tag}/ex']), wpe=kw.get('wpe', True),
            wpe_taps=kw.get('wpe_tabs', 10), wpe_delay=kw.get('wpe_delay', 2),
            wpe_iterations=kw.get('wpe_iterations', 3), stft_size=size, stft_shift=shift,
            bss_iterations=kw['bss_iterations'],
            bss_iterations_post=kw.get('bss_iterations_post', 1),
            bf_drop_context=kw.get('bf_drop_context', True),
            bf=kw.get('bf', 'mvdrSouden_ban'), postfilter=kw.get('postfilter'),
            return_details=True)
        sel = slice(None, None, 16) if fixture.endswith('1024.npz') else slice(None)
        if f'{tag}/acitivity_freq' in g.files:
            assert np.array_equal(det['activity_freq'], g[f'{tag}/acitivity_freq']), tag
        if f'{tag}/context_frames' in g.files:
            assert (det['start_context_frames'], det['end_context_frames']) == \
                tuple(g[f'{tag}/context_frames']), tag
        assert np.max(np.abs(det['masks'][..., sel] - g[f'{tag}/masks'])) < 1e-9, tag
        assert np.array_equal(det['masks'][..., sel] == 0, g[f'{tag}/masks'] == 0), tag
        assert rel_err(det['X_hat'][..., sel], g[f'{tag}/X_hat']) < 1e-9, tag
        assert rel_err(x_hat, g[f'{tag}/x_hat']) < 1e-9, tag


def test_gss_initialization_fixture(golden):
    g = golden('orchestration_small.npz')
    init, mask = oracle.gss_initialization(g['default/acitivity_freq'])
    assert np.array_equal(init, g['default/gss_initialization'])
    assert np.array_equal(mask, g['default/gss_source_active_mask'])


def test_oracle_beamformer_fixture(golden):
    g = golden('beamformer.npz')
    Y = g['Y']
    X2, N2 = np.median(g['X_mask3'], axis=0), np.median(g['N_mask3'], axis=0)
    got, det = oracle.beamform_mvdr_souden_from_masks(Y, X2, N2, ban=True, return_details=True)
    assert rel_err(got, g['ban_2d']) < 1e-12
    assert rel_err(det['cov_x'], g['Cov_X']) < 1e-12
    assert rel_err(det['cov_n'], g['Cov_N']) < 1e-12
    assert rel_err(det['w'], g['w_souden_ban']) < 1e-12
    # singular distortion PSD in one bin: lstsq fallback gives a zero filter there
    w = oracle.get_mvdr_vector_souden(
        oracle.get_power_spectral_density_matrix(Y.transpose(2, 0, 1), X2.T),
        oracle.get_power_spectral_density_matrix(Y.transpose(2, 0, 1), g['N_mask_zero_bin'].T),
        eps=1e-10)
    assert rel_err(w, g['w_souden_zero_bin']) < 1e-12
    assert not w[3].any()


def test_windows_of_product_equal_oracle():
    from pb_chime5_amd import ops
    for size, shift in ((1024, 256), (512, 128), (64, 16), (512, 256)):
        a = ops.analysis_window(size)
        assert np.array_equal(a, oracle.blackman_periodic(size))
        assert np.array_equal(ops.synthesis_window(a, shift), oracle.biorthogonal_window(a, shift))


def test_context_bookkeeping_fixture():
    """backup_orig_start_end / adjust_start_end / AddContext against the reference's own
    functions run on CHiME-5 shaped examples (database.py:540-570, 706-1053)."""
    import copy
    from conftest import GOLDEN
    from pb_chime5_amd.database.chime5.context import (
        AddContext, adjust_start_end, backup_orig_start_end)
    cases = json.loads((GOLDEN / 'context_bookkeeping.json').read_text())
    assert len(cases) == 6
    for name, case in cases.items():
        ex = copy.deepcopy(case['input'])
        ex = backup_orig_start_end(ex)
        if case['adjust']:
            ex = adjust_start_end(ex)
        samples = case['samples']
        samples = tuple(samples) if isinstance(samples, list) else samples
        ex = AddContext(samples, equal_start_context=case['equal'])(ex)
        for key, want in case['output'].items():
            assert ex[key] == want, (name, key)
    with pytest.raises(AssertionError):
        AddContext(-1)
