"""End-to-end parity of the fused device pipeline and of the reference-shaped Python
surface (get_enhancer / Enhancer / blocks) against

* the golden fixtures captured from the reference's own orchestration code
  (tests/golden/make_golden.py), and
* the CPU oracle on the seeded synthetic configs of BASELINE.json.

Bar (north star): <= 1e-4 relative on the enhanced STFT magnitude, bit-exact for
activity / indexing.  The asserts below use the tighter bounds actually met.
"""
import json
import os

import numpy as np
import pytest

import gss_oracle as oracle
from conftest import rel_err, rel_err_per_freq

pytestmark = pytest.mark.gpu

TOL_STFT_MAG = 1e-4     # the north star's tolerance on |X_hat|


def _enhancer_for(case_kwargs, stft):
    from pb_chime5_amd.core import get_enhancer
    return get_enhancer(stft_size=int(stft[0]), stft_shift=int(stft[1]), **case_kwargs)


def _ex(vec):
    return {'start': {'original': int(vec[0])}, 'start_orig': {'original': int(vec[1])},
            'end_orig': {'original': int(vec[2])}, 'end': {'original': int(vec[3])}}


def _cases(g):
    return sorted({k.split('/')[0] for k in g.files})


def _inputs(g, tag):
    src = tag if f'{tag}/obs' in g.files else 'default'
    act = g[f'{src}/activity']
    names = [f'P{k + 1:02d}' for k in range(act.shape[0] - 1)] + ['Noise']
    return g[f'{src}/obs'], dict(zip(names, act))


@pytest.mark.parametrize('fixture', ['orchestration_small.npz', 'orchestration_1024.npz'])
def test_reference_orchestration_fixtures(gpu_ctx, golden, fixture):
    g = golden(fixture)
    for tag in _cases(g):
        kwargs = json.loads(str(g[f'{tag}/kwargs']))
        obs, activity = _inputs(g, tag)
        speaker = str(g[f'{tag}/speaker'])
        enh = _enhancer_for(kwargs, g[f'{tag}/stft'])
        x_hat = enh.enhance_observation(obs, activity, speaker, ex=_ex(g[f'{tag}/ex']),
                                        debug=True)
        loc = enh.enhance_observation_locals
        want_x = g[f'{tag}/x_hat']
        assert x_hat.shape == want_x.shape and x_hat.dtype == np.float64, tag
        # integer / bool: exact
        assert loc['target_speaker_index'] == int(g[f'{tag}/target_speaker_index']), tag
        if f'{tag}/acitivity_freq' in g.files:
            assert np.array_equal(loc['acitivity_freq'], g[f'{tag}/acitivity_freq']), tag
        if f'{tag}/context_frames' in g.files:
            assert (loc['start_context_frames'], loc['end_context_frames']) == \
                tuple(g[f'{tag}/context_frames']), tag
        sel = slice(None, None, 16) if fixture.endswith('1024.npz') else slice(None)
        # float: enhanced STFT magnitude within the north-star tolerance
        X_want = g[f'{tag}/X_hat']
        X_got = loc['X_hat'][..., sel]
        assert rel_err(np.abs(X_got), np.abs(X_want)) < TOL_STFT_MAG, tag
        assert rel_err(X_got, X_want) < 1e-6, tag
        assert np.max(np.abs(loc['target_mask'][..., sel] - g[f'{tag}/target_mask'])) < 1e-6, tag
        assert np.max(np.abs(loc['distortion_mask'][..., sel]
                             - g[f'{tag}/distortion_mask'])) < 1e-6, tag
        assert np.max(np.abs(loc['masks'][..., sel] - g[f'{tag}/masks'])) < 1e-6, tag
        if not fixture.endswith('1024.npz'):
            assert rel_err(x_hat, want_x) < 1e-6, tag
        if f'{tag}/Obs' in g.files:
            assert rel_err(loc['Obs'][..., sel], g[f'{tag}/Obs']) < 1e-8, tag
        # zeroed context frames are exactly zero
        zero = g[f'{tag}/target_mask'] == 0
        assert np.all(loc['target_mask'][..., sel][zero] == 0), tag


def test_block_by_block_path_equals_fused(gpu_ctx, golden):
    g = golden('orchestration_small.npz')
    kwargs = json.loads(str(g['default/kwargs']))
    obs, activity = _inputs(g, 'default')
    enh = _enhancer_for(kwargs, g['default/stft'])
    ex = _ex(g['default/ex'])
    a = enh.enhance_observation(obs, activity, 'P01', ex=ex, fused=True)
    b = enh.enhance_observation(obs, activity, 'P01', ex=ex, fused=False, debug=True)
    assert rel_err(a, b) < 1e-10
    assert rel_err(b, g['default/x_hat']) < 1e-6
    assert enh.gss_block.locals['initialization'].shape == g['default/gss_initialization'].shape
    assert np.array_equal(enh.gss_block.locals['initialization'],
                          g['default/gss_initialization'])
    assert np.array_equal(enh.gss_block.locals['source_active_mask'],
                          g['default/gss_source_active_mask'])


def test_wpe_block_4d_fixture(gpu_ctx, golden):
    from pb_chime5_amd.core import WPE
    g = golden('wpe_block.npz')
    w = WPE(taps=2, delay=1, iterations=2, psd_context=0)
    assert rel_err(w(g['Obs'], stack=True), g['stack_true']) < 1e-9
    assert rel_err(w(g['Obs'], stack=False), g['stack_false']) < 1e-9
    assert rel_err(w(g['Obs'][0]), g['ndim3']) < 1e-9
    with pytest.raises(NotImplementedError):
        w(g['Obs'], stack='x')
    with pytest.raises(NotImplementedError):
        w(g['Obs'][0, 0])


def test_enhance_example_fixture(gpu_ctx, golden, monkeypatch):
    """Reference Enhancer.enhance_example: channel selection, interval -> dense
    activity, common-length cut, context trim (core.py:396-512)."""
    from pb_chime5_amd import core
    from pb_chime5_amd.utils.intervall_array import ArrayIntervall
    g = golden('enhance_example.npz')
    ex = json.loads(str(g['ex']))
    arrays = sorted(ex['audio_path']['observation'])
    audio = {a: g[f'audio/{a}'] for a in arrays}
    monkeypatch.setattr(core, 'load_audio',
                        lambda path, start=None, stop=None, dtype=None: audio[path][:, start:stop])
    total = audio['U01'].shape[1]
    store = {'S99': {}}
    for a in arrays:
        store['S99'][a] = {}
        for s in g['speakers']:
            ai = ArrayIntervall(total + 500)
            if str(s) == 'Noise':
                ai[0:total + 500] = 1
            else:
                for lo, hi in g[f'intervals/{s}']:
                    ai[int(lo):int(hi)] = 1
            store['S99'][a][str(s)] = ai
    for tag, multiarray in [('true', True), ('outer', 'outer_array_mics'),
                            ('first', 'first_array_mics'), ('false', False)]:
        enh = core.get_enhancer(multiarray=multiarray, context_samples=400, wpe=True,
                                wpe_tabs=2, wpe_iterations=1, bss_iterations=2,
                                stft_size=64, stft_shift=16, activity_store=store)
        x_hat = enh.enhance_example(ex, debug=True)
        loc = enh.enhance_example_locals
        assert np.array_equal(loc['obs'], g[f'{tag}/obs']), tag          # indexing: exact
        assert np.array_equal(np.array(list(loc['ex_array_activity'].values())),
                              g[f'{tag}/activity']), tag
        assert x_hat.shape == g[f'{tag}/x_hat'].shape, tag
        assert rel_err(x_hat, g[f'{tag}/x_hat']) < 1e-6, tag


def test_error_types_match_reference(gpu_ctx):
    from pb_chime5_amd.core import get_enhancer
    from pb_chime5_amd import synthetic
    u = synthetic.tiny()
    enh = get_enhancer(bf='nope', wpe=False, bss_iterations=1)
    with pytest.raises(NotImplementedError):
        enh.enhance_observation(u.obs, u.activity, u.speaker_id, ex=u.ex)
    enh = get_enhancer(postfilter='nope', wpe=False, bss_iterations=1)
    with pytest.raises(NotImplementedError):
        enh.enhance_observation(u.obs, u.activity, u.speaker_id, ex=u.ex)
    enh = get_enhancer(wpe=False, bss_iterations=1)
    bad = dict(u.ex, start_orig={'original': -5})
    with pytest.raises(AssertionError):
        enh.enhance_observation(u.obs, u.activity, u.speaker_id, ex=bad)
    with pytest.raises(ValueError):
        enh.enhance_observation(u.obs, u.activity, 'P99', ex=u.ex)
    with pytest.raises(AssertionError):
        get_enhancer(wpe=1)
    with pytest.raises(RuntimeError):
        enh.enhance_example({'session_id': 'S02', 'speaker_id': 'P05'})


# ---------------------------------------------------------------- synthetic configs
def test_config1_full_pipeline_vs_oracle(gpu_ctx):
    """BASELINE.json configs[0]: 4 mics, 5 s, 2 speakers, WPE off, 5 EM iterations."""
    from pb_chime5_amd import ops, synthetic
    u = synthetic.config1(context=16000)
    x_hat, det = ops.enhance_observation(
        u.obs, u.activity_array, u.target_index, 16000, 16000, wpe=False, bss_iterations=5,
        debug=True, ctx=gpu_ctx)
    want, wdet = oracle.enhance_observation(
        u.obs, u.activity_array, u.target_index, u.ex, wpe=False, bss_iterations=5,
        return_details=True, gss_fn=oracle.gss_block_batched)
    assert np.array_equal(det['acitivity_freq'], wdet['activity_freq'])
    assert det['ref_channel'] == wdet['ref_channel']
    assert (wdet['start_context_frames'], wdet['end_context_frames']) == (66, 66)
    assert rel_err(np.abs(det['X_hat']), np.abs(wdet['X_hat'])) < TOL_STFT_MAG
    assert rel_err(det['X_hat'], wdet['X_hat']) < 1e-6
    assert np.max(np.abs(det['target_mask'] - wdet['target_mask'])) < 1e-6
    assert x_hat.shape == want.shape == (80128,)
    assert rel_err(x_hat, want) < 1e-6


def test_tiny_with_wpe_vs_oracle(gpu_ctx):
    from pb_chime5_amd import ops, synthetic
    u = synthetic.tiny(num_channels=6, num_samples=24000, num_speakers=3, context=4096)
    x_hat, det = ops.enhance_observation(
        u.obs, u.activity_array, u.target_index, 4096, 4096, wpe=True, wpe_taps=5,
        bss_iterations=8, debug=True, ctx=gpu_ctx)
    want, wdet = oracle.enhance_observation(
        u.obs, u.activity_array, u.target_index, u.ex, wpe=True, wpe_taps=5, bss_iterations=8,
        return_details=True, gss_fn=oracle.gss_block_batched)
    assert det['ref_channel'] == wdet['ref_channel']
    assert rel_err(det['Obs'], wdet['Obs']) < 1e-6      # cond(R) * eps, see test_gpu_stages
    assert rel_err(np.abs(det['X_hat']), np.abs(wdet['X_hat'])) < TOL_STFT_MAG
    assert rel_err(x_hat, want) < 1e-4


@pytest.fixture(scope='module')
def config2_run(gpu_ctx):
    from pb_chime5_amd import ops, synthetic
    u = synthetic.config2()
    ctx_samples = u.ex['start_orig']['original']
    x_hat, det = ops.enhance_observation(
        u.obs, u.activity_array, u.target_index, ctx_samples, ctx_samples, debug=True,
        ctx=gpu_ctx)
    return u, x_hat, det


def test_config2_stagewise_vs_oracle_on_frequency_subset(gpu_ctx, config2_run):
    """BASELINE.json configs[1] at full size (24 ch, 15 s, WPE 10 taps, 20 EM
    iterations), stage by stage: each stage is checked on a subset of frequency bins,
    feeding the oracle the GPU's own upstream tensors (so that every stage is held to
    its own, much tighter, bound); the last stage is checked on all bins."""
    from pb_chime5_amd import ops
    u, x_hat, det = config2_run
    bins = [3, 40, 129, 300, 511]
    assert det['Obs'].shape == (24, 941, 513)
    # STFT + WPE on the subset
    Y = oracle.stft(u.obs)[..., bins]
    X_want = oracle.wpe_block(Y, 10, 2, 3)
    assert rel_err(det['Obs'][..., bins], X_want) < 1e-7
    # activity: exact
    act_f = oracle.activity_time_to_frequency(u.activity_array, 1024, 256, True)
    assert np.array_equal(det['acitivity_freq'], act_f)
    # EM on the subset, from the GPU's dereverberated tensor
    post_want = oracle.gss_block(det['Obs'][..., bins], act_f, 20, 1)
    post_got = det['posterior'][..., bins]
    assert np.max(np.abs(post_got - post_want)) < 1e-5
    assert np.max(np.abs(det['posterior'].sum(axis=0) - 1)) < 1e-12
    # masks + beamformer on all bins, from the GPU's posteriors
    masks = det['posterior'].copy()
    sf, ef = oracle.start_end_context_frames(u.ex, 1024, 256, True)
    masks[:, :sf] = 0
    masks[:, -ef:] = 0
    tm = masks[u.target_index]
    dm = np.sum(np.delete(masks, u.target_index, axis=0), axis=0)
    assert np.array_equal(det['target_mask'], tm)
    assert np.max(np.abs(det['distortion_mask'] - dm)) < 1e-15
    X_hat_want, bdet = oracle.beamform_mvdr_souden_from_masks(
        det['Obs'], tm, dm, ban=True, return_details=True)
    assert det['ref_channel'] == bdet['ref_channel']
    assert rel_err(np.abs(det['X_hat']), np.abs(X_hat_want)) < TOL_STFT_MAG
    assert rel_err(det['X_hat'], X_hat_want) < 1e-7
    assert rel_err(x_hat, oracle.istft(det['X_hat'])) < 1e-11


def _scene_all_bins(gpu_ctx, pool, u, *, bf='mvdrSouden_ban', bss_iterations=20):
    """A SURVEY 8d scene at full size, ALL 513 bins, no bin left out: dereverberated tensor
    and posteriors against the oracle run end to end (its WPE and EM spread over worker
    processes), the beamformer stage through _beamformer_all_bins_with_referee."""
    from pb_chime5_amd import ops
    cs = u.ex['start_orig']['original']
    ce = u.ex['end']['original'] - u.ex['end_orig']['original']
    x_hat, det = ops.enhance_observation(u.obs, u.activity_array, u.target_index, cs, ce,
                                         debug=True, ctx=gpu_ctx, bf=bf,
                                         bss_iterations=bss_iterations)
    D, T = u.obs.shape[0], det['Obs'].shape[1]
    assert det['Obs'].shape == (D, T, 513)
    _, wdet = oracle.enhance_observation(
        u.obs, u.activity_array, u.target_index, u.ex, return_details=True, bf=bf,
        bss_iterations=bss_iterations, gss_fn=pool.gss_block, wpe_fn=pool.wpe_block)
    assert np.array_equal(det['acitivity_freq'], wdet['activity_freq'][:, :T])
    nrm = np.linalg.norm(wdet['Obs'], axis=(0, 1))
    wpe_err = np.linalg.norm(det['Obs'] - wdet['Obs'], axis=(0, 1)) / nrm
    post_err = np.max(np.abs(det['posterior'] - np.where(wdet['masks'] == 0, det['posterior'],
                                                         wdet['masks'])), axis=(0, 1))
    print('all bins: after WPE per-bin error max %.2e (bin %d) median %.2e; posteriors max %.2e '
          '(bin %d) median %.2e' % (wpe_err.max(), int(np.argmax(wpe_err)), np.median(wpe_err),
                                    post_err.max(), int(np.argmax(post_err)), np.median(post_err)))
    assert wpe_err.max() < 1e-6
    assert post_err.max() < 1e-4
    masks = det['posterior'].copy()
    sf, ef = oracle.start_end_context_frames(u.ex, 1024, 256, True)
    masks[:, :sf] = 0
    if ef > 0:
        masks[:, -ef:] = 0
    assert np.array_equal(det['target_mask'], masks[u.target_index])
    _beamformer_all_bins_with_referee(gpu_ctx, det, u, bf)
    assert rel_err(x_hat, oracle.istft(det['X_hat'])) < 1e-11
    assert np.all(np.isfinite(x_hat))
    return x_hat, det


def _beamformer_all_bins_with_referee(gpu_ctx, det, u, bf):
    """The beamformer stage on ALL 513 bins of a SURVEY 8d scene, from the GPU's own
    dereverberated tensor and masks.

    On these scenes (point sources, sensor noise 60 dB down) the distortion PSD matrix is
    nearly singular in most bins (config 3: cond(Phi_N) > 1e8 in 480 of 513 bins, > 1e12 in
    100), and there the REFERENCE'S OWN float64 arithmetic does not evaluate its own
    formulas: np.linalg.solve loses cond * eps, blind_analytic_normalization's four-operand
    einsum cancels catastrophically once cond^2 * eps > 1, and the cross-frequency SNR sums
    of the reference-channel search come out as 1e60.  Measured against an 80-bit evaluation
    of the same formulas (tests/ext_precision.py) the literal float64 oracle is off by
    5e-4 (median) for cond in [1e8, 1e10), by 0.7 for [1e10, 1e12) and by factors of 1e2 ...
    1e5 beyond, while a float64 evaluation in the stable order ||Phi_N w|| / |w^H Phi_N w|
    -- what mvdr_apply_kernel computes -- stays within 0.2 cond eps of it in every bin.

    So every bin is held to a bound, none is left out:
      * literal oracle, 1e-4, wherever the reference's float64 result is itself meaningful
        (cond(Phi_N) < 1e8);
      * everywhere: the extended-precision evaluation of the reference's formulas, within
        max(1e-6, 4 cond eps) for MVDR (the error bound of a backward-stable float64 solve;
        = 1e-4 at cond 1e11).
    The reference channel is an argmax over SNR sums that the degenerate bins dominate: it
    must equal the oracle's unless the oracle's own SNRs are rounding noise (> 1e12), in
    which case the oracle's channel is forced for the comparison."""
    from pb_chime5_amd import ops
    import ext_precision as xp
    eps = np.finfo(np.float64).eps
    tm, dm = det['target_mask'], det['distortion_mask']
    Yf = det['Obs'].transpose(2, 0, 1)
    cov_x = oracle.get_power_spectral_density_matrix(Yf, tm.T)
    cov_n = oracle.get_power_spectral_density_matrix(Yf, dm.T)
    cond = np.linalg.cond(cov_n)
    strict = cond < 1e8
    X_got = det['X_hat']
    ref = None
    if bf == 'gev_ban':
        X_lit = oracle.beamform_gev_from_masks(det['Obs'], tm, dm, ban=True)
        c_bound = 200.0       # an eigenvector also feels the eigenvalue gap
    else:
        phi = oracle.stable_solve(cov_n, cov_x)
        mat = phi / np.maximum(np.trace(phi, axis1=-1, axis2=-2)[..., None, None].real, 1e-10)
        num = np.einsum('FdR,FdD,FDR->R', mat.conj(), cov_x, mat).real
        den = np.einsum('FdR,FdD,FDR->R', mat.conj(), cov_n, mat).real
        snr = num / np.maximum(den, 1e-10)
        ref = int(np.argmax(snr))
        print('oracle SNR max %.3g, reference channel oracle %d gpu %d' % (snr.max(), ref, det['ref_channel']))
        if det['ref_channel'] != ref:
            assert snr.max() > 1e12, (det['ref_channel'], ref, snr.max())
            X_got = ops.mvdr_souden_from_masks(det['Obs'], tm, dm, ban=True, ref_channel=ref,
                                               ctx=gpu_ctx)
        w = oracle.get_mvdr_vector_souden(cov_x, cov_n, ref_channel=ref, eps=1e-10)
        X_lit = oracle.apply_beamforming_vector(oracle.blind_analytic_normalization(w, cov_n), Yf).T
        c_bound = 4.0
    X_ext = xp.beamformer_all_bins(det['Obs'], tm, dm, ref, bf, workers=12)
    n = lambda a: np.linalg.norm(a, axis=0)
    scale = n(np.abs(X_ext))
    e_ext = n(np.abs(X_got) - np.abs(X_ext)) / scale
    e_lit = n(np.abs(X_got) - np.abs(X_lit)) / scale
    o_ext = n(np.abs(X_lit) - np.abs(X_ext)) / scale
    tol = np.maximum(1e-6, c_bound * cond * eps)
    print('cond(Phi_N) < 1e8 in %d bins, < 1e10 in %d, < 1e12 in %d of %d'
          % (strict.sum(), (cond < 1e10).sum(), (cond < 1e12).sum(), cond.size))
    print('GPU vs extended precision: max of err / (cond eps) %.3f; bins where the bound exceeds '
          '1e-4: %d; literal oracle vs extended precision: %d bins above 1e-4, max %.2e'
          % (np.max(e_ext / (cond * eps)), (tol > 1e-4).sum(), (o_ext > 1e-4).sum(), o_ext.max()))
    assert strict.any()
    assert np.all(e_lit[strict] < TOL_STFT_MAG), np.flatnonzero(strict & (e_lit >= TOL_STFT_MAG))
    assert np.all(e_ext < tol), [(int(f), e_ext[f], tol[f]) for f in np.flatnonzero(e_ext >= tol)[:8]]


def test_config3_dev_shaped_utterance_all_bins(gpu_ctx, oracle_pool):
    """BASELINE.json configs[2] as SURVEY 8d specifies it: a dev-shaped utterance (24 ch,
    reference-default context of 240000 samples on both sides -> about 2000 frames), every
    stage on all 513 bins."""
    from pb_chime5_amd import synthetic
    u = synthetic.config3_item(0)
    x_hat, det = _scene_all_bins(gpu_ctx, oracle_pool, u)
    assert det['Obs'].shape[1] > 1900


def test_config5_long_rttm_segment_gev_all_bins(gpu_ctx, oracle_pool):
    """BASELINE.json configs[4] as SURVEY 8d specifies it: 120 s, 12 channels
    ('outer_array_mics'), 40 EM iterations, GEV + BAN beamformer (T = 7503 frames: the STFT
    tensor alone is 740 MB), every stage on all 513 bins."""
    from pb_chime5_amd import synthetic
    u = synthetic.config5()
    x_hat, det = _scene_all_bins(gpu_ctx, oracle_pool, u, bf='gev_ban', bss_iterations=40)
    assert det['Obs'].shape[:2] == (12, 7503)


def _all_bins_vs_oracle(gpu_ctx, pool, u, *, bf='mvdrSouden_ban', bss_iterations=20):
    """The whole utterance against the LITERAL oracle on ALL 513 bins (WPE and EM of the
    oracle spread over worker processes), the oracle's own reference channel, no bin left
    out: for scenes whose distortion PSD matrix is well conditioned in every bin."""
    from pb_chime5_amd import ops
    cs = u.ex['start_orig']['original']
    ce = u.ex['end']['original'] - u.ex['end_orig']['original']
    x_hat, det = ops.enhance_observation(u.obs, u.activity_array, u.target_index, cs, ce,
                                         debug=True, ctx=gpu_ctx, bf=bf,
                                         bss_iterations=bss_iterations)
    want, wdet = oracle.enhance_observation(
        u.obs, u.activity_array, u.target_index, u.ex, return_details=True, bf=bf,
        bss_iterations=bss_iterations, gss_fn=pool.gss_block, wpe_fn=pool.wpe_block)
    T = det['Obs'].shape[1]
    assert np.array_equal(det['acitivity_freq'], wdet['activity_freq'][:, :T])
    Yf = wdet['Obs'].transpose(2, 0, 1)
    cond = np.linalg.cond(oracle.get_power_spectral_density_matrix(Yf, wdet['distortion_mask'].T))
    print('cond(Phi_N): max %.3g median %.3g' % (cond.max(), np.median(cond)))
    assert cond.max() < 1e6, cond.max()
    errs = dict(
        wpe=rel_err(det['Obs'], wdet['Obs']),
        wpe_per_f=rel_err_per_freq(det['Obs'], wdet['Obs'], axis=-1),
        masks=float(np.max(np.abs(det['posterior'] - np.where(
            wdet['masks'] == 0, det['posterior'], wdet['masks'])))),
        X_mag=rel_err(np.abs(det['X_hat']), np.abs(wdet['X_hat'])),
        X_mag_per_f=rel_err_per_freq(np.abs(det['X_hat']), np.abs(wdet['X_hat']), axis=-1),
        x=rel_err(x_hat, want))
    print(errs)
    if bf == 'mvdrSouden_ban':
        assert det['ref_channel'] == wdet['ref_channel']
        assert rel_err(det['X_hat'], wdet['X_hat']) < TOL_STFT_MAG
        assert errs['x'] < TOL_STFT_MAG
    assert errs['wpe'] < 1e-6 and errs['wpe_per_f'] < 1e-5
    assert errs['masks'] < 1e-3, errs['masks']
    assert errs['X_mag'] < TOL_STFT_MAG and errs['X_mag_per_f'] < TOL_STFT_MAG
    return det, wdet


def test_config3_all_bins_literal_oracle_well_conditioned(gpu_ctx, oracle_pool):
    """BASELINE configs[2] shape (24 ch, 34.7 s incl. 2 x 15 s context, T = 2169) with a
    spatially diffuse background 6 dB below one talker, so that cond(Phi_N) < 1e6 in every
    bin: all bins, literal oracle, oracle-chosen reference channel."""
    from pb_chime5_amd import synthetic
    n = 554490
    iv = [(240000, n - 240000), (100000, 400000), (50000, 250000), (300000, 520000)]
    u = synthetic.make_utterance(1000, 24, n, iv, start_context=240000, end_context=240000,
                                 rir_taps=1024, noise=0.1, fast=True, diffuse_noise=0.5)
    det, _ = _all_bins_vs_oracle(gpu_ctx, oracle_pool, u)
    assert det['Obs'].shape == (24, 2169, 513)


def test_config5_all_bins_literal_oracle_well_conditioned(gpu_ctx, oracle_pool):
    """BASELINE configs[4] shape (120 s, 12 ch, T = 7503, 40 EM iterations, GEV + BAN) with the
    same diffuse background: all bins against the literal oracle (the phase of a generalised
    eigenvector is arbitrary upstream too, so magnitudes are compared)."""
    from pb_chime5_amd import synthetic
    sr = 16000
    iv = [(50 * sr, 70 * sr), (10 * sr, 60 * sr), (40 * sr, 100 * sr), (65 * sr, 115 * sr)]
    u = synthetic.make_utterance(5, 12, 120 * sr, iv, start_context=50 * sr, end_context=50 * sr,
                                 noise=0.1, fast=True, diffuse_noise=0.5)
    det, _ = _all_bins_vs_oracle(gpu_ctx, oracle_pool, u, bf='gev_ban', bss_iterations=40)
    assert det['Obs'].shape == (12, 7503, 513)


def test_fused_pipeline_with_wpe_psd_context(gpu_ctx):
    """get_enhancer(wpe_psd_context=2) stays on the fused device pipeline and matches the
    oracle; the block-by-block path agrees."""
    from pb_chime5_amd import synthetic
    from pb_chime5_amd.core import get_enhancer
    u = synthetic.tiny(seed=21, num_channels=6, num_samples=24000, num_speakers=3, context=4096)
    enh = get_enhancer(wpe_tabs=4, wpe_psd_context=2, bss_iterations=6)
    assert enh._fusable()
    got = enh.enhance_observation(u.obs, u.activity, u.speaker_id, ex=u.ex)
    blocks = enh.enhance_observation(u.obs, u.activity, u.speaker_id, ex=u.ex, fused=False)
    want = oracle.enhance_observation(u.obs, u.activity_array, u.target_index, u.ex, wpe_taps=4,
                                      wpe_psd_context=2, bss_iterations=6,
                                      gss_fn=oracle.gss_block_batched)
    plain = oracle.enhance_observation(u.obs, u.activity_array, u.target_index, u.ex, wpe_taps=4,
                                       bss_iterations=6, gss_fn=oracle.gss_block_batched)
    assert rel_err(got, want) < 1e-5 and rel_err(blocks, got) < 1e-9
    assert rel_err(plain, want) > 1e-3


def test_config2_end_to_end_vs_oracle(gpu_ctx, config2_run, oracle_pool):
    """The bench workload (BASELINE.json configs[1], the SURVEY 8d generator) end to end
    against the oracle run on ALL 513 bins.  Measured: 3e-8 after WPE, 1e-5 (global scale) on
    the enhanced signal -- the 20 EM iterations amplify the last-bit differences of the WPE
    solve.

    Per frequency bin the picture is less uniform: on this scene (point sources, sensor noise
    60 dB down) the guided EM is, in a few bins, sensitive enough to its input that the
    ORACLE'S OWN output moves by more than 1e-4 when the observation is perturbed in the last
    bit (bin 169: 3e-4).  So every bin is held to 1e-4 on its own scale unless the oracle
    itself is not reproducible to that level there: the tolerance of a bin is
    max(1e-4, 100 x the oracle's last-bit sensitivity in that bin), the bins that need the
    second term must stay below 3 %, and no bin may be off by more than 1e-2.  The
    well-conditioned variant below has no such bins.

    Round 3 (WPE panels by 16-blocked triangular substitution instead of the explicit
    48 x 48 inverse): 1 of 513 bins above 1e-4 (bin 169: 2.9e-4 at an oracle
    self-sensitivity of 8e-5), so the second term is now 10 x, at most 2 bins may need it,
    and the global figure is held to 3e-5."""
    u, x_hat, det = config2_run
    kw = dict(return_details=True, gss_fn=oracle_pool.gss_block, wpe_fn=oracle_pool.wpe_block)
    want, wdet = oracle.enhance_observation(u.obs, u.activity_array, u.target_index, u.ex, **kw)
    assert rel_err(det['Obs'], wdet['Obs']) < 1e-6
    assert det['ref_channel'] == wdet['ref_channel']
    glob = rel_err(np.abs(det['X_hat']), np.abs(wdet['X_hat']))
    print('config 2 global |X_hat| error %.2e, x_hat %.2e, after WPE %.2e'
          % (glob, rel_err(x_hat, want), rel_err(det['Obs'], wdet['Obs'])))
    assert glob < 3e-5
    assert rel_err(x_hat, want) < 3e-5
    # the oracle against itself, observation perturbed in the last bit
    rng = np.random.default_rng(0)
    obs2 = u.obs * (1 + 2e-16 * rng.standard_normal(u.obs.shape))
    _, wdet2 = oracle.enhance_observation(obs2, u.activity_array, u.target_index, u.ex, **kw)
    A, B, B2 = np.abs(det['X_hat']), np.abs(wdet['X_hat']), np.abs(wdet2['X_hat'])
    nb = np.linalg.norm(B, axis=0)
    loud = nb ** 2 > np.max(nb ** 2) * 1e-8                    # within 80 dB of the loudest bin
    err = np.linalg.norm(A - B, axis=0) / nb
    self_f = np.linalg.norm(B2 - B, axis=0) / nb
    tol = np.maximum(TOL_STFT_MAG, 10.0 * self_f)
    relaxed = loud & (err >= TOL_STFT_MAG)                     # bins that need the second term
    print('config 2 per-bin |X_hat| error: median %.2e, max %.2e (bin %d, oracle self-sensitivity '
          '%.2e there); %d of %d bins above 1e-4; oracle sensitivity above 1e-6 in %d bins'
          % (np.median(err[loud]), err[loud].max(), int(np.argmax(np.where(loud, err, 0))),
             self_f[int(np.argmax(np.where(loud, err, 0)))], int((err[loud] > TOL_STFT_MAG).sum()),
             int(loud.sum()), int((loud & (self_f > 1e-6)).sum())))
    assert loud.all()
    assert np.all(err[loud] < tol[loud]), np.flatnonzero(loud & (err >= tol))
    assert relaxed.sum() <= 2 and err[loud].max() < 1e-3
    # the WPE stage itself, per bin: GPU vs oracle against the oracle's own last-bit noise
    wpe_err = np.linalg.norm(det['Obs'] - wdet['Obs'], axis=(0, 1))
    wpe_self = np.linalg.norm(wdet2['Obs'] - wdet['Obs'], axis=(0, 1))
    ratio = wpe_err / np.maximum(wpe_self, 1e-300)
    print('WPE output, GPU - oracle over oracle self-noise per bin: median %.2f, max %.2f (bin %d)'
          % (np.median(ratio), ratio.max(), int(np.argmax(ratio))))
    # (measured: median 1.98, max 2.66 -- the correlation matrix is summed frame by frame on
    # the MFMA, BLAS sums in blocks; test_wpe_config2_bins_within_oracle_noise_... has the
    # same factor against the extended-precision solution)
    assert np.median(ratio) < 3.0 and ratio.max() < 5.0


def test_config2_well_conditioned_all_bins_per_frequency(gpu_ctx, oracle_pool):
    """The config-2 shape (24 ch, 15 s, T = 941, K = 5) with a spatially diffuse background
    6 dB below one talker: the guided EM is then well posed in every bin and EVERY bin
    meets 1e-4 on its own scale (measured: 1e-8), literal oracle, oracle's reference
    channel."""
    from pb_chime5_amd import synthetic
    n = 240000
    iv = [(70000, 170000), (20000, 130000), (90000, 230000), (0, 110000)]
    u = synthetic.make_utterance(2, 24, n, iv, start_context=80000, end_context=80000,
                                 noise=0.1, fast=True, diffuse_noise=0.5)
    det, _ = _all_bins_vs_oracle(gpu_ctx, oracle_pool, u)
    assert det['Obs'].shape == (24, 941, 513)


def test_config2_properties(gpu_ctx, config2_run):
    """Size-independent properties at full size: determinism and exact
    equivariance to a power-of-two input gain (masks are scale invariant, every
    linear stage scales exactly)."""
    from pb_chime5_amd import ops
    u, x_hat, det = config2_run
    ctx_samples = u.ex['start_orig']['original']
    again = ops.enhance_observation(u.obs, u.activity_array, u.target_index, ctx_samples,
                                    ctx_samples, ctx=gpu_ctx)
    assert np.array_equal(again, x_hat)
    scaled = ops.enhance_observation(4.0 * u.obs, u.activity_array, u.target_index, ctx_samples,
                                     ctx_samples, ctx=gpu_ctx)
    assert rel_err(scaled, 4.0 * x_hat) < 1e-9
    assert np.all(np.isfinite(x_hat))
    # the target is enhanced relative to the interferers: output power in the
    # target-only region vs. the region where the target is silent
    assert np.std(x_hat) > 0


# ---------------------------------------------------------------- reference channel
def _oracle_snr(cov_x, cov_n):
    """The per-channel SNR get_optimal_reference_channel maximises, from the oracle's PSDs."""
    phi = oracle.stable_solve(cov_n, cov_x)
    mat = phi / np.maximum(np.trace(phi, axis1=-1, axis2=-2)[..., None, None].real, 1e-10)
    num = np.einsum('FdR,FdD,FDR->R', mat.conj(), cov_x, mat).real
    den = np.einsum('FdR,FdD,FDR->R', mat.conj(), cov_n, mat).real
    return num / np.maximum(den, 1e-10)


def _check_ref_channel_or_tie(gpu_ctx, det, wdet, tag, mismatches, well_conditioned=None,
                              postfilter=None):
    """The reference channel is one integer and must equal the oracle's.  A different
    channel is tolerated only when it is CERTIFIED that the oracle's own answer is not
    defined to that precision:
      * a tie: the oracle's SNRs of the two candidates agree to 1e-9 relative, or
      * a degenerate scene: the oracle's SNR of one of the two candidates moves by more than
        their gap when the noise PSD matrices are perturbed in the LAST BIT (a noise PSD
        matrix that is singular to rounding: the SNR sums are decided by the rounding of the
        solve in the reference as well).
    Even then the beamformer is still checked, with the oracle's channel forced
    (gss_mvdr_souden_ref) on the GPU's own tensors.  Every mismatch is recorded in the
    session-wide list `mismatches`, whose fixture bounds them at teardown."""
    from pb_chime5_amd import ops
    if det['ref_channel'] == wdet['ref_channel']:
        return True
    g, o = det['ref_channel'], wdet['ref_channel']
    snr = _oracle_snr(wdet['cov_x'], wdet['cov_n'])
    gap = abs(snr[g] - snr[o]) / abs(snr[o])
    rng = np.random.default_rng(0)
    jitter = 1 + 2.2e-16 * rng.standard_normal(wdet['cov_n'].shape)
    snr2 = _oracle_snr(wdet['cov_x'], wdet['cov_n'] * jitter)
    moved = max(abs(snr2[g] - snr[g]) / abs(snr[g]), abs(snr2[o] - snr[o]) / abs(snr[o]))
    cond = np.linalg.cond(wdet['cov_n'])
    mismatches.append(dict(tag=tag, gpu=g, oracle=o, gap=float(gap), last_bit_move=float(moved),
                           cond_max=float(cond.max())))
    assert gap < 1e-9 or moved > gap or not np.isfinite(moved), (tag, g, o, gap, moved, cond.max())
    if well_conditioned is not None and gap >= 1e-9:
        # (an exact tie -- two classes the EM cannot tell apart end with equal masks,
        # Phi_X = Phi_N and the same SNR on every channel -- is a tie on any scene)
        assert not well_conditioned, (tag, 'reference channel differs on a well-conditioned scene')
    X_forced = ops.mvdr_souden_from_masks(det['Obs'], det['target_mask'], det['distortion_mask'],
                                          ban=True, ref_channel=o, ctx=gpu_ctx)
    if postfilter == 'mask_mul':
        X_forced = X_forced * det['target_mask']
    good = cond < 1e8
    assert good.any(), tag
    assert rel_err(np.abs(X_forced[:, good]), np.abs(wdet['X_hat'][:, good])) < TOL_STFT_MAG, tag
    return False


# ---------------------------------------------------------------- edge cases
def _run_both(u, **kw):
    from pb_chime5_amd import ops
    cs = u.ex['start_orig']['original']
    ce = u.ex['end']['original'] - u.ex['end_orig']['original']
    got, det = ops.enhance_observation(u.obs, u.activity_array, u.target_index, cs, ce,
                                       debug=True, **kw)
    want, wdet = oracle.enhance_observation(u.obs, u.activity_array, u.target_index, u.ex,
                                            return_details=True,
                                            gss_fn=oracle.gss_block_batched, **kw)
    return got, det, want, wdet


def test_edge_very_short_utterance(gpu_ctx):
    """Fewer frames than channels: every class covariance is rank deficient, the
    1e-10 eigenvalue floor decides (eigendecomposition path)."""
    from pb_chime5_amd import synthetic
    u = synthetic.make_utterance(3, 4, 700, [(100, 600), (0, 400)], rir_taps=64)
    got, det, want, wdet = _run_both(u, wpe=False, bss_iterations=4)
    assert det['Obs'].shape[1] == 6 == wdet['Obs'].shape[1]
    assert got.shape == want.shape
    assert np.array_equal(det['acitivity_freq'], wdet['activity_freq'])
    assert np.max(np.abs(det['posterior'] - np.where(wdet['masks'] == 0, det['posterior'],
                                                      wdet['masks']))) < 1e-4


def test_edge_target_never_active_and_silent_speaker(gpu_ctx):
    from pb_chime5_amd import synthetic
    u = synthetic.tiny(seed=5, num_channels=4, num_samples=9000, num_speakers=3, context=1024)
    u.activity['P01'][:] = False          # target has no activity at all
    u.activity['P03'][:] = False
    got, det, want, wdet = _run_both(u, wpe=True, wpe_taps=3, bss_iterations=4)
    assert det['ref_channel'] == wdet['ref_channel']
    assert rel_err(np.abs(det['X_hat']), np.abs(wdet['X_hat'])) < TOL_STFT_MAG
    assert rel_err(got, want) < 1e-4


def test_edge_dead_microphone_takes_the_lstsq_branch(gpu_ctx):
    """One microphone delivers digital zeros: its STFT must be EXACT zeros so that the WPE
    correlation matrix is exactly singular, np.linalg.solve raises in the reference and
    stable_solve falls back to the minimum-norm lstsq solution (math/solve.py:95-114) --
    zero taps on the dead channel.  (A two-for-one real FFT leaks 1e-16 of the partner
    channel; the STFT kernel zeroes all-zero frames explicitly.)"""
    from pb_chime5_amd import synthetic
    u = synthetic.tiny(seed=3, num_channels=6, num_samples=24000, num_speakers=2)
    u.obs[2] = 0.0
    got, det, want, wdet = _run_both(u, wpe=True, wpe_taps=4, wpe_delay=2, wpe_iterations=2,
                                     bss_iterations=5)
    assert np.all(det['Obs'][2] == 0) and np.all(wdet['Obs'][2] == 0)
    assert rel_err(det['Obs'], wdet['Obs']) < 1e-5
    assert rel_err(np.abs(det['X_hat']), np.abs(wdet['X_hat'])) < TOL_STFT_MAG
    assert rel_err(got, want) < TOL_STFT_MAG


@pytest.mark.parametrize('channels,wpe', [(4, False), (4, True), (6, False), (6, True)])
def test_edge_block_of_digital_silence_in_every_channel(gpu_ctx, channels, wpe):
    """Every microphone delivers digital zeros for a while (a dropped block): those STFT frames
    are exact zeros, their quadratic forms sit on the clamp max(|q|, tiny), and THERE the
    posterior depends on the eigenvalue normalisation of pb_bss (lambda / lambda_max) -- the one
    place where the Cholesky form of the model update (B_k up to a scale) is not equivalent.
    Frequencies that hold such frames take the eigendecomposition for every class
    (em_prepare_kernel flags them; found by the zero-frame test of the stage in round 6)."""
    from pb_chime5_amd import synthetic
    u = synthetic.tiny(seed=11, num_channels=channels, num_samples=30000, num_speakers=3, context=2048)
    u.obs[:, 9000:17000] = 0.0
    got, det, want, wdet = _run_both(u, wpe=wpe, wpe_taps=3, wpe_delay=2, wpe_iterations=2,
                                     bss_iterations=5)
    zero = np.all(wdet['Obs'] == 0, axis=0)                      # (T, F)
    assert zero.all(axis=1).sum() >= 20                          # whole frames of zeros reach the EM
    assert det['ref_channel'] == wdet['ref_channel']
    # (the reference's masks have their context frames zeroed; elsewhere they are the posteriors)
    post, masks = det['posterior'], wdet['masks']
    assert post.shape == masks.shape
    inside = masks.sum(axis=0) > 0
    assert inside[zero].sum() > 20 * 100                         # zero frames outside the context
    # (without the exact path the zero frames differ by 0.1 ... 1; behind WPE the first frames of
    # the silence hold the filter's prediction alone, whose direction carries the 1e-6 between
    # the two solvers)
    assert np.max(np.abs(post - masks)[:, inside]) < (1e-4 if wpe else 1e-6)
    assert np.max(np.abs(post - masks)[:, inside & zero]) < (1e-4 if wpe else 1e-6)
    assert rel_err(np.abs(det['X_hat']), np.abs(wdet['X_hat'])) < TOL_STFT_MAG
    assert rel_err(got, want) < TOL_STFT_MAG


def test_edge_all_zero_observation_gives_nan_like_reference(gpu_ctx):
    """Digital silence: PSD matrices are zero, solve falls back to lstsq -> w = 0, and
    BAN divides 0 / 0 (eps = 0 upstream), so the reference returns NaN everywhere.
    (With WPE switched on the reference does not get that far: 1 / max(0, 0) = inf
    poisons R and np.linalg.eigh raises; the GPU path raises as well, at the next place
    the reference checks for finiteness.)"""
    from pb_chime5_amd import ops, synthetic
    u = synthetic.tiny(num_channels=3, num_samples=6000, num_speakers=2, context=512)
    u.obs[:] = 0
    got, det, want, wdet = _run_both(u, wpe=False, bss_iterations=2)
    assert np.all(np.isnan(want)) and np.all(np.isnan(got))
    # with WPE the NaN reaches the SNRs of the reference-channel search, which the reference
    # asserts to be finite (had it come that far)
    with pytest.raises(AssertionError):
        ops.enhance_observation(u.obs, u.activity_array, u.target_index, 512, 512,
                                wpe=True, wpe_taps=2, bss_iterations=2)


def test_edge_single_class_and_two_channels(gpu_ctx):
    from pb_chime5_amd import ops
    rng = np.random.default_rng(12)
    obs = rng.standard_normal((2, 5000))
    act = np.ones((1, 5000), bool)
    ex = {'start': {'original': 0}, 'start_orig': {'original': 0},
          'end_orig': {'original': 5000}, 'end': {'original': 5000}}
    got, det = ops.enhance_observation(obs, act, 0, 0, 0, wpe=True, wpe_taps=2,
                                       bss_iterations=2, debug=True)
    want, wdet = oracle.enhance_observation(obs, act, 0, ex, wpe_taps=2, bss_iterations=2,
                                            return_details=True,
                                            gss_fn=oracle.gss_block_batched)
    # one class: posteriors are exactly 1, the distortion mask is 0 -> lstsq path -> NaN
    assert np.all(det['posterior'] == 1.0) and np.all(wdet['masks'][0][3:-3] == 1.0)
    assert np.array_equal(np.isnan(got), np.isnan(want))


@pytest.mark.parametrize('D,K', [(20, 4), (28, 3), (12, 6), (24, 2), (16, 3), (10, 5), (4, 3), (6, 4)])
def test_other_channel_and_class_counts(gpu_ctx, ref_mismatches, D, K):
    """Channel counts off the config-2 path: D = 28 / 16 / 6 take the LDS-form E-step and
    other MFMA tile counts (wpe_apply with 1 or 2 channel tiles, 3 or 4 x 8 lanes in the
    class update), D = 12 / 24 with K = 6 / 2 the register-form E-step at its limits,
    D = 20 / 10 / 4 its other instantiations (5-array sessions, one array); D <= 12 the
    one-wave-per-sub-tile correlation, D = 4 / 6 with per-wave staging, D = 4 the
    register-form M-step."""
    from pb_chime5_amd import synthetic
    # enough frames per unknown (T = 254, taps * D <= 56) and sensor noise 30 dB below the
    # speech (with 1 - 3 point sources on 12 - 28 microphones the spatial covariance is
    # otherwise numerically singular) for a well-conditioned WPE
    u = synthetic.tiny(seed=D + K, num_channels=D, num_samples=64000, num_speakers=K - 1,
                       context=4096, noise=3e-2)
    got, det, want, wdet = _run_both(u, wpe=True, wpe_taps=2, wpe_delay=2, wpe_iterations=2,
                                     bss_iterations=6)
    assert np.array_equal(det['acitivity_freq'], wdet['activity_freq'])
    assert rel_err(det['Obs'], wdet['Obs']) < 1e-5
    _check_ref_channel_or_tie(gpu_ctx, det, wdet, (D, K), ref_mismatches)
    # one point source on 24 microphones (D, K = 24, 2) leaves the noise PSD matrix singular
    # to rounding in most bins (median cond 3e11, max 3e18): there the reference's own
    # output is decided by rounding, so the beamformer is compared where cond(Phi_N) < 1e8
    # (see _stagewise) and the time signal only when that is every bin
    strict = np.linalg.cond(wdet['cov_n']) < 1e8
    assert strict.mean() > 0.05, strict.mean()
    if det['ref_channel'] != wdet['ref_channel']:
        return          # a certified tie: X_hat was compared with the oracle's channel forced
    assert rel_err(np.abs(det['X_hat'][:, strict]), np.abs(wdet['X_hat'][:, strict])) < TOL_STFT_MAG
    if strict.all():
        assert rel_err(got, want) < TOL_STFT_MAG
    else:
        assert (D, K) == (24, 2), (D, K, strict.mean())


@pytest.mark.parametrize('D', [4, 12, 24])
def test_kernel_variants_agree(gpu_ctx, D, monkeypatch):
    """The tuned kernel variants against their plain counterparts on the same input: 16 x 16
    vs 32 x 32 correlation tiles, 3- vs 4-product complex MFMA forms (correlation and filter
    application), the filter application with 1 - 4 frame phases packed into the column
    dimension, register-form vs tiled M-step (D = 4), register- vs LDS-form E-step,
    Cholesky vs eigendecomposition model update, chunked vs statically partitioned M-step, the
    one-launch EM of one array vs three launches per iteration, the correlation kernel's forms,
    the EM over blocks of frequencies on one and two streams.
    Same arithmetic up to summation order."""
    from pb_chime5_amd import ops, synthetic
    # (frames per unknown and sensor noise as in test_other_channel_and_class_counts: a
    # well-conditioned WPE, so that summation order does not decide the result)
    u = synthetic.tiny(seed=40 + D, num_channels=D, num_samples=64000, num_speakers=2,
                       context=4096, noise=3e-2)
    cs = u.ex['start_orig']['original']
    ce = u.ex['end']['original'] - u.ex['end_orig']['original']

    def run():
        return ops.enhance_observation(u.obs, u.activity_array, u.target_index, cs, ce, wpe=True,
                                       wpe_taps=2, wpe_iterations=2, bss_iterations=5,
                                       debug=True)
    base, bdet = run()
    # the beamformer is compared where the noise PSD matrix is not singular to rounding
    # (few point sources on many microphones, see test_other_channel_and_class_counts)
    dm = np.sum(np.delete(bdet['posterior'], u.target_index, axis=0), axis=0)
    cond = np.linalg.cond(oracle.get_power_spectral_density_matrix(bdet['Obs'].transpose(2, 0, 1),
                                                                   dm.T))
    good = cond < 1e8
    assert good.mean() > 0.05, good.mean()
    for variant in ('corr_ts=2' if D <= 12 else 'corr_ts=1', 'corr_ts=1', 'mstep_tiled', 'estep_lds',
                    'force_eigh', 'corr_nw=2', 'chol_diag_unfolded',
                    'apply_ph=1', 'apply_ph=2', 'apply_ph=3', 'apply_ph=4',
                    # M-step in chunks instead of the static partition, the one-array EM as
                    # separate launches, block-wise accumulation and pair tiles in the
                    # correlation, the general form of the filter application; two at once
                    'mstep_chunked', 'mstep_slots=100', 'em_unfused', 'corr_blocked', 'corr_ts=3', 'mstep_generic',
                    'apply_generic', 'corr_p_tiles', 'estep_lds,force_eigh,corr_ts=3',
                    # one array: single waves over all frames / two waves splitting them (default
                    # four), 8 staging registers per lane
                    'corr_ksplit=1', 'corr_ksplit=2', 'corr_ksplit=1,corr_stg8', 'corr_stg8',
                    # 24 channels: two frame phases in three full column tiles with G from global
                    # memory instead of the unpacked filter application
                    'apply_gglobal',
                    # the EM over blocks of frequencies (long segments: Infinity-Cache
                    # residency), one and two blocks in flight, more segments per frequency
                    'em_l3_fit_mb=0,em_l3_mb=1,em_streams=1', 'em_l3_fit_mb=0,em_l3_mb=1,em_streams=2',
                    'em_l3_fit_mb=0,em_l3_mb=2,em_streams=2,mstep_maxseg=16,estep_lds'):
        env = {'GSS_VARIANT': variant}
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        other, odet = run()
        for k in env:
            monkeypatch.delenv(k)
        errs = (rel_err(odet['Obs'], bdet['Obs']), np.max(np.abs(odet['posterior'] - bdet['posterior'])),
                rel_err(odet['X_hat'][:, good], bdet['X_hat'][:, good]))
        print(D, env, errs)
        # the WPE solve amplifies rounding by cond(R) (1.6e-6 against the oracle in the worst
        # bin of this kind of scene), the EM by another two orders of magnitude
        assert errs[0] < 1e-5 and errs[1] < 1e-3 and errs[2] < TOL_STFT_MAG, (env, errs)
        if good.all():
            assert rel_err(other, base) < TOL_STFT_MAG, (env, rel_err(other, base))


def test_pipeline_with_a_window_that_is_not_a_power_of_two(gpu_ctx):
    """get_enhancer(stft_size=400, stft_shift=100) (core.py:577-579 takes any length): the whole
    pipeline on the direct-DFT STFT / iSTFT kernels against the oracle."""
    from pb_chime5_amd import synthetic
    from pb_chime5_amd.core import get_enhancer
    u = synthetic.tiny(seed=33, num_channels=5, num_samples=20000, num_speakers=2, context=2000)
    enh = get_enhancer(stft_size=400, stft_shift=100, wpe_tabs=3, bss_iterations=4)
    got = enh.enhance_observation(u.obs, u.activity, u.speaker_id, ex=u.ex)
    want = oracle.enhance_observation(u.obs, u.activity_array, u.target_index, u.ex, wpe_taps=3,
                                      bss_iterations=4, stft_size=400, stft_shift=100,
                                      gss_fn=oracle.gss_block_batched)
    assert got.shape == want.shape
    assert rel_err(got, want) < 1e-5


def test_unsupported_sizes_fail_loudly(gpu_ctx):
    from pb_chime5_amd import ops
    # pb_bss CACGMMTrainer.fit: assert K < 20 -- an AssertionError in the reference too
    with pytest.raises(AssertionError):
        ops.enhance_observation(np.zeros((2, 4000)), np.ones((20, 4000), bool), 0, 0, 0)
    with pytest.raises(NotImplementedError):
        ops.stft(np.zeros(4000), size=15, shift=5)          # odd window length
    with pytest.raises(NotImplementedError):
        ops.stft(np.zeros(40000), size=8192, shift=2048)
    with pytest.raises(ValueError):
        ops.enhance_observation(np.zeros((2, 4000)), np.ones((2, 100), bool), 0, 0, 0)
    # 2^31 STFT bins or more (24 channels: 48 minutes in one piece; a 15-minute utterance runs,
    # tools/long_utterance_check.py): refused before anything is touched
    from ctypes import c_void_p
    buf = gpu_ctx.empty(64)
    T = (1 << 31) // (513 * 24) + 1
    with pytest.raises(NotImplementedError, match='STFT bins'):
        gpu_ctx._check(gpu_ctx.lib.gss_wpe(gpu_ctx.handle, c_void_p(buf.ptr), 513, T, 24, 10, 2, 3,
                                           0, c_void_p(buf.ptr + 32)), 'gss_wpe')


def test_utterance_pipeline_is_bit_identical_to_one_at_a_time(gpu_ctx):
    """ops.UtterancePipeline (several utterances in flight on separate HIP streams, slot
    buffers reused across utterances of different sizes) returns exactly what the
    one-at-a-time path returns, in submission order."""
    from pb_chime5_amd import ops, synthetic
    utts = [synthetic.tiny(seed=s, num_channels=c, num_samples=n, num_speakers=k)
            for s, c, n, k in [(1, 4, 12000, 2), (2, 6, 20000, 3), (3, 4, 9000, 2),
                               (4, 8, 16000, 2), (5, 6, 20000, 3)]]
    kw = dict(wpe=True, wpe_taps=3, wpe_iterations=2, bss_iterations=4)
    params = ops.make_params(**kw)
    want = []
    for u in utts:
        cs = u.ex['start_orig']['original']
        want.append(ops.enhance_observation(u.obs, u.activity_array, u.target_index, cs, cs,
                                            params=params, ctx=gpu_ctx))
    for depth in (1, 2, 3):
        pipe = ops.UtterancePipeline(params, depth=depth, first_ctx=gpu_ctx)
        got = {}
        for i, u in enumerate(utts):
            cs = u.ex['start_orig']['original']
            if pipe.full():
                tag, x = pipe.pop()
                got[tag] = x
            pipe.enqueue(i, u.obs, u.activity_array, u.target_index, cs, cs)
        while len(pipe):
            tag, x = pipe.pop()
            got[tag] = x
        pipe.close()
        assert sorted(got) == list(range(len(utts)))
        for i, w in enumerate(want):
            assert np.array_equal(got[i], w), (depth, i)


def test_pcm16_input_is_bit_identical_to_float64(gpu_ctx):
    """gss_enhance_observation_pcm16: int16 samples converted inside the STFT kernel
    (x * 2^-15, what the reference's loader does on the host) -- same bits out."""
    from pb_chime5_amd import ops, synthetic
    u = synthetic.tiny(seed=9, num_channels=6, num_samples=20000, num_speakers=3)
    pcm = np.clip(np.rint(u.obs / np.abs(u.obs).max() * 20000), -32768, 32767).astype(np.int16)
    as_float = pcm.astype(np.float64) / 2 ** 15
    cs = u.ex['start_orig']['original']
    params = ops.make_params(wpe=True, wpe_taps=3, wpe_iterations=2, bss_iterations=4)
    want = ops.enhance_observation(as_float, u.activity_array, u.target_index, cs, cs,
                                   params=params, ctx=gpu_ctx)
    pipe = ops.UtterancePipeline(params, depth=2, first_ctx=gpu_ctx)
    pipe.enqueue('pcm', pcm, u.activity_array, u.target_index, cs, cs)
    pipe.enqueue('f64', as_float, u.activity_array, u.target_index, cs, cs)
    got = dict(pipe.pop() for _ in range(2))
    pipe.close()
    assert np.array_equal(got['f64'], want)
    assert np.array_equal(got['pcm'], want)


def _wpe_stage_ok(det, wdet, taps, delay, iterations, kw=None):
    """The WPE outputs agree to 1e-6, or -- normal equations with cond(R) ~ 1e12 put BOTH float64
    results that far from the exact least-squares solution -- the GPU is no further from the
    extended-precision iteration (tests/ext_precision.py) than 3 x the oracle, over the four
    frequencies where the two differ most."""
    import ext_precision
    Xg, Xo = det['Obs'], wdet['Obs']
    if rel_err(Xg, Xo) < 1e-6:
        return True
    kw = kw or {}
    if kw.get('wpe_psd_context', 0) != 0 or 'obs_in' not in det:
        return False        # (the extended-precision iteration is written for psd_context = 0)
    Y = oracle.stft(det['obs_in'], kw.get('stft_size', 1024), kw.get('stft_shift', 256),
                    fading=kw.get('stft_fading', True))
    n = np.linalg.norm
    per = np.array([n(Xg[..., f] - Xo[..., f]) / max(n(Xo[..., f]), 1e-300)
                    for f in range(Xg.shape[-1])])
    dg, do = [], []
    for f in np.argsort(per)[::-1][:4]:
        Yf = np.ascontiguousarray(Y[..., f])
        Xt = ext_precision.wpe(Yf, oracle.build_y_tilde(Yf, taps, delay), iterations)[-1]
        dg.append(n(Xg[..., f] - Xt) / n(Xt))
        do.append(n(Xo[..., f] - Xt) / n(Xt))
    # (frequency by frequency the two scatter by a factor of ten at cond(R) = 1e12: the worst
    # of the four against the worst of the four, and no single one beyond 10 x its partner)
    return max(dg) <= 3 * max(max(do), 1e-9) and all(g <= 10 * max(o, 1e-9) for g, o in zip(dg, do))


def _fuzz_case(gpu_ctx, ref_mismatches, case, D, K, N, ctx_s, kw, wide=False, mutate=None):
    from pb_chime5_amd import ops, synthetic
    u = synthetic.tiny(seed=5000 + case, num_channels=D, num_samples=N, num_speakers=K - 1,
                       context=ctx_s, noise=5e-2)
    if mutate is not None:
        mutate(u)            # tools/fuzz_silence.py: blocks of digital silence in the recording
    bf = kw['bf']
    tag = (case, D, K, N, kw)
    raised = []
    try:
        got, det = ops.enhance_observation(u.obs, u.activity_array, u.target_index, ctx_s, ctx_s,
                                           debug=True, ctx=gpu_ctx, **kw)
    except (np.linalg.LinAlgError, AssertionError) as e:
        raised.append('gpu ' + type(e).__name__)
    try:
        want, wdet = oracle.enhance_observation(u.obs, u.activity_array, u.target_index, u.ex,
                                                return_details=True,
                                                gss_fn=oracle.gss_block_batched, **kw)
    except (np.linalg.LinAlgError, AssertionError) as e:
        raised.append('oracle ' + type(e).__name__)
    if raised:
        # The reference aborts the utterance: GEV with a noise PSD matrix that is not positive
        # definite (LinAlgError from scipy.linalg.eigh), MVDR with an SNR that is not finite
        # (the assert in get_optimal_reference_channel).  Both are decided by the rounding of
        # a matrix that is singular to working precision, so one side raising alone is not an
        # error -- but it has to be the exception of that beamformer.
        kinds = {r.split()[1] for r in raised}
        assert kinds == {'LinAlgError' if bf == 'gev_ban' else 'AssertionError'} and \
            bf in ('gev_ban', 'mvdrSouden_ban'), (tag, raised)
        return 'raises: ' + ', '.join(raised)
    assert got.shape == want.shape, tag
    assert np.array_equal(det['acitivity_freq'], wdet['activity_freq'][:, :det['Obs'].shape[1]]), tag
    det['obs_in'] = u.obs
    if wide:
        # the reference's block-by-block orchestration (WPE, GSS, Beamformer objects, core.py)
        # over the stage operators must give what the fused pipeline gives
        from pb_chime5_amd.core import get_enhancer
        enh = get_enhancer(wpe=kw['wpe'], wpe_tabs=kw['wpe_taps'], wpe_delay=kw['wpe_delay'],
                           wpe_iterations=kw['wpe_iterations'],
                           wpe_psd_context=kw.get('wpe_psd_context', 0),
                           stft_size=kw.get('stft_size', 1024), stft_shift=kw.get('stft_shift', 256),
                           stft_fading=kw.get('stft_fading', True),
                           bss_iterations=kw['bss_iterations'],
                           bss_iterations_post=kw['bss_iterations_post'], bf=bf,
                           postfilter=kw['postfilter'])
        blocks = enh.enhance_observation(u.obs, u.activity, u.speaker_id, ex=u.ex, fused=False)
        assert blocks.shape == got.shape, tag
        assert np.array_equal(np.isnan(blocks), np.isnan(got)), (tag, 'block path vs fused: NaN pattern')
        if bf != 'gev_ban' and not np.isnan(got).any():     # (a generalised eigenvector's phase: two kernels, two answers)
            scale = max(np.abs(got).max(), 1e-300)
            assert np.max(np.abs(blocks - got)) < 1e-9 * scale, \
                (tag, 'block path vs fused', np.max(np.abs(blocks - got)) / scale)
    if wide and kw.get('wpe_psd_context', 0) and 1e-6 <= rel_err(det['Obs'], wdet['Obs']) < 1e-4:
        return 'ill-conditioned WPE (psd context: no referee)'
    assert _wpe_stage_ok(det, wdet, kw['wpe_taps'], kw['wpe_delay'], kw['wpe_iterations'], kw), \
        (tag, rel_err(det['Obs'], wdet['Obs']))
    if rel_err(det['Obs'], wdet['Obs']) >= 1e-6:
        return 'ill-conditioned WPE'    # both results rounding-decided: nothing downstream compares
    # the masks: equal to 1e-6, or -- classes left with fewer effective frames than channels
    # have a continuum of eigenvalues that the model's 1e-10 floor cuts through, and any two
    # float64 implementations then disagree -- on the frequency that differs most the GPU is no
    # further from the extended-precision EM of tests/ext_precision.py than 5 x the oracle is
    if wide and np.abs(wdet['X_hat']).max() < 1e-30 * np.abs(wdet['Obs']).max():
        # a target whose posterior never rises above e.g. 1e-170 (17 frames, 26 channels): the
        # mask-multiplied output is numerically zero and its relative error that of exp(-400)
        assert np.abs(det['X_hat']).max() < 1e-25 * np.abs(wdet['Obs']).max(), tag
        return 'silent target'

    def per_bin(a, b):
        return np.linalg.norm(a - b, axis=0) / np.maximum(np.linalg.norm(b, axis=0), 1e-300)
    loud = np.linalg.norm(wdet['X_hat'], axis=0)
    loud = loud > 1e-4 * loud.max()                    # bins that matter for the output
    dmask = np.maximum(per_bin(det['target_mask'], wdet['target_mask']),
                       per_bin(det['distortion_mask'], wdet['distortion_mask'])) * loud
    if dmask.max() > 1e-6:
        import ext_precision
        f = int(np.argmax(dmask))
        act = wdet['activity_freq'][:, :wdet['Obs'].shape[1]]
        Of = np.ascontiguousarray(wdet['Obs'][..., f:f + 1])
        it, post = kw['bss_iterations'], kw['bss_iterations_post']
        g = ops.cacgmm_posteriors(Of, act, it, post, ctx=gpu_ctx)[..., 0]
        o = oracle.gss_block_batched(Of, act, iterations=it, iterations_post=post)[..., 0]
        # the referee: the same EM in 80-bit extended precision (tests/ext_precision.py); the GPU
        # may be no further from it than 5 x what the reference's own arithmetic leaves undecided
        # (oracle - referee, or the oracle's movement under last-bit input changes)
        r = ext_precision.guided_em(np.ascontiguousarray(Of[..., 0].T), act, it, post)
        d_gr = np.max(np.abs(g - r))
        yard = ext_precision.em_yardstick(Of, act, it, post, o, r)
        assert d_gr <= 5 * yard + 1e-9, (tag, "EM of frequency", f, d_gr, yard)
        return 'sensitive EM'
    if bf == 'mvdrSouden_ban':
        # bins with a nearly singular Phi_N are decided by rounding in the reference too
        cond = np.linalg.cond(wdet['cov_n'])
        good = cond < 1e8
        if wide and good.mean() <= 0.5:
            return 'singular Phi_N'     # one point source on many microphones (K = 2)
        assert good.mean() > 0.5, tag
        if _check_ref_channel_or_tie(gpu_ctx, det, wdet, tag, ref_mismatches,
                                     well_conditioned=bool(cond.max() < 1e8),
                                     postfilter=kw['postfilter']):
            if rel_err(np.abs(det['X_hat'][:, good]), np.abs(wdet['X_hat'][:, good])) >= TOL_STFT_MAG:
                # blind_analytic_normalization's einsum loses cond^2 eps in the reference (see
                # _beamformer_all_bins_with_referee): the frequencies that differ go to the
                # extended-precision evaluation of the same formulas, bound 4 cond eps
                import ext_precision as xp
                per = np.linalg.norm(np.abs(det['X_hat']) - np.abs(wdet['X_hat']), axis=0) / \
                    np.linalg.norm(wdet['X_hat'][:, good])
                eps = np.finfo(np.float64).eps
                for f in np.flatnonzero(good & (per >= 0.03 * TOL_STFT_MAG)):
                    ref = xp.mvdr_souden_ban_output(det['Obs'][..., f], det['target_mask'][:, f],
                                                    det['distortion_mask'][:, f], det['ref_channel'])
                    if kw['postfilter'] == 'mask_mul':
                        ref = ref * det['target_mask'][:, f]
                    err = np.linalg.norm(np.abs(det['X_hat'][:, f]) - np.abs(ref)) / np.linalg.norm(ref)
                    assert err < max(1e-6, 4 * cond[f] * eps), (tag, int(f), err, cond[f])
    elif bf == 'gev_ban':
        # the principal eigenvector's phase is arbitrary per frequency: magnitudes, where
        # the generalised eigenproblem is not singular to rounding
        # ... and the principal eigenvalue is separated from the next (two classes that the
        # EM cannot tell apart end with equal masks, Phi_X = Phi_N and every vector a solution)
        Yf = wdet['Obs'].transpose(2, 0, 1)
        cov_n = oracle.get_power_spectral_density_matrix(Yf, det['distortion_mask'].T)
        cov_x = oracle.get_power_spectral_density_matrix(Yf, det['target_mask'].T)
        good = np.linalg.cond(cov_n) < 1e8
        lam = np.sort(np.linalg.eigvals(np.linalg.solve(cov_n[good], cov_x[good])).real, axis=-1)
        sep = np.zeros(len(good), bool)
        sep[good] = lam[:, -1] - lam[:, -2] > 1e-3 * np.abs(lam[:, -1])
        good &= sep
        if wide and good.mean() <= 0.5:
            return 'degenerate GEV'
        assert good.mean() > 0.5, tag
        assert rel_err(np.abs(det['X_hat'][:, good]), np.abs(wdet['X_hat'][:, good])) \
            < TOL_STFT_MAG, tag
    else:
        assert rel_err(np.abs(det['X_hat']), np.abs(wdet['X_hat'])) < TOL_STFT_MAG, tag
        assert rel_err(got, want) < TOL_STFT_MAG, tag


def test_random_shapes_against_oracle(gpu_ctx, ref_mismatches):
    """Seeded fuzz over channel / class / frame counts, context, WPE and EM settings, beamformer
    and postfilter: every stage has partial-tile
    and odd-size code paths that the BASELINE shapes never visit."""
    # GSS_FUZZ_SEED / GSS_FUZZ_CASES / GSS_FUZZ_WIDE: the same sweep from another seed, longer,
    # over up to 12 classes and the GEV beamformer, all failures collected (bug hunts outside
    # the suite; tools/fuzz_case.py replays one case)
    import fuzz_params
    seed, cases, wide = fuzz_params.from_environment()
    done, failures, notes = 0, [], {}
    for case, D, K, N, ctx_s, kw in fuzz_params.fuzz_cases(seed, cases, wide):
        if wide:
            try:
                note = _fuzz_case(gpu_ctx, ref_mismatches, case, D, K, N, ctx_s, kw, wide=True)
                notes[note] = notes.get(note, 0) + 1
            except AssertionError as e:
                failures.append(str(e)[:600])
        else:
            _fuzz_case(gpu_ctx, ref_mismatches, case, D, K, N, ctx_s, kw)
        done += 1
    if wide:
        print('fuzz:', done, 'cases;', notes)
    assert not failures, '\n'.join(failures)
    assert done >= min(15, cases // 3), done


def test_wpe_in_two_sets_of_frequencies_gives_the_same_bits(gpu_ctx, monkeypatch):
    """gss_set_utterances_in_flight (ABI 6): told that there is one utterance at a time on the GPU
    (Enhancer.enhance_observation says so) the fused pipeline runs the WPE stage as two sets of frequencies side by side on
    the context's stream and an internal second one (a set's solve under the other's
    correlation).  Frequencies are independent: every tap of the pipeline is BIT-identical to
    the one-stream run (hint 2 or 0, or GSS_VARIANT=wpe_halves=0), for an even and an uneven split,
    the zeroed-pivot count of an underdetermined WPE still reaches the status word, and the
    context is usable on one stream again afterwards."""
    from pb_chime5_amd import ops, synthetic
    u = synthetic.tiny(seed=61, num_channels=6, num_samples=48000, num_speakers=2, context=4096,
                       noise=3e-2)
    cs = u.ex['start_orig']['original']
    ce = u.ex['end']['original'] - u.ex['end_orig']['original']

    def run(**kw):
        return ops.enhance_observation(u.obs, u.activity_array, u.target_index, cs, ce, wpe=True,
                                       wpe_taps=3, wpe_iterations=2, bss_iterations=4, debug=True,
                                       ctx=gpu_ctx, **kw)
    gpu_ctx.set_utterances_in_flight(2)
    one, d_one = run()
    gpu_ctx.set_utterances_in_flight(1)
    two, d_two = run()
    monkeypatch.setenv('GSS_VARIANT', 'wpe_halves=20')
    uneven, d_uneven = run()
    monkeypatch.setenv('GSS_VARIANT', 'wpe_halves=0')
    off, d_off = run()
    monkeypatch.delenv('GSS_VARIANT')
    for other, det in ((two, d_two), (uneven, d_uneven), (off, d_off)):
        assert np.array_equal(other, one)
        for key in ('Obs', 'posterior', 'X_hat', 'target_mask'):
            assert np.array_equal(det[key], d_one[key]), key
    assert gpu_ctx.last_wpe_zero_pivots() == 0
    # fewer frames than unknowns: pivots are zeroed in both sets and counted once
    short = synthetic.tiny(seed=62, num_channels=8, num_samples=9000, num_speakers=2, context=1000)
    counts = []
    for hint in (2, 1):
        gpu_ctx.set_utterances_in_flight(hint)
        ops.enhance_observation(short.obs, short.activity_array, short.target_index, 1000, 1000,
                                wpe=True, wpe_taps=10, wpe_iterations=1, bss_iterations=2,
                                ctx=gpu_ctx)
        counts.append(gpu_ctx.last_wpe_zero_pivots())
    assert counts[0] == counts[1] > 0, counts
    with pytest.raises(ValueError):
        gpu_ctx.set_utterances_in_flight(-1)
    gpu_ctx.set_utterances_in_flight(0)


def test_adopted_stream_and_profiling_with_the_second_stream(gpu_ctx):
    """gss_set_stream + gss_profile_*: the fused pipeline on a stream the caller owns,
    with the per-kernel profile switched on, one stream and two (gss_set_utterances_in_flight 1):
    the same bits as on the context's own stream; the profile of the two-stream run lists twice
    the WPE launches (two sets of frequencies) and the same EM launches; the caller's stream is
    what the result is ordered on (no synchronisation of ours in between)."""
    import ctypes
    from pb_chime5_amd import ops, synthetic
    u = synthetic.tiny(seed=71, num_channels=5, num_samples=30000, num_speakers=2, context=3000)
    kw = dict(wpe=True, wpe_taps=3, wpe_iterations=2, bss_iterations=3, ctx=gpu_ctx)

    def run():
        return ops.enhance_observation(u.obs, u.activity_array, u.target_index, 3000, 3000, **kw)
    want = run()
    hip = ctypes.CDLL('libamdhip64.so')          # a stream of the caller's own
    stream = ctypes.c_void_p()
    assert hip.hipStreamCreate(ctypes.byref(stream)) == 0
    gpu_ctx.set_stream(stream.value)
    try:
        calls = {}
        for hint in (0, 1):
            gpu_ctx.set_utterances_in_flight(hint)
            gpu_ctx.profile_enable(True)
            gpu_ctx.profile_reset()
            got = run()
            prof = gpu_ctx.profile_report()
            gpu_ctx.profile_enable(False)
            assert np.array_equal(got, want), hint
            calls[hint] = {k: v['calls'] for k, v in prof.items()}
        assert calls[1]['wpe_corr'] == 2 * calls[0]['wpe_corr'] == 4
        assert calls[1]['wpe_apply'] == 2 * calls[0]['wpe_apply']
        assert calls[1]['em_estep'] == calls[0]['em_estep'] and calls[1]['stft'] == calls[0]['stft'] == 1
    finally:
        gpu_ctx.set_utterances_in_flight(0)
        gpu_ctx.set_stream(None)
        hip.hipStreamDestroy(stream)
    assert np.array_equal(run(), want)
