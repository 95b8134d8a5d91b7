"""Stage goldens from the REAL third-party packages, for the day they are installable:

    pip install nara_wpe  &&  git submodule update --init   # pb_bss
    python tests/golden/make_golden_upstream.py

writes tests/golden/upstream_stages.npz.  The reference takes every float on the hot path
from nara_wpe (>= 0.0.6, /root/reference/setup.py:142) and pb_bss (/root/reference/.gitmodules:1-3);
neither is in this image (no network, not in the offline wheelhouse, the submodule directory is
empty), so today this script stops at the import and the consuming tests
(tests/test_upstream_goldens.py) skip with that reason.  Until then the oracle is pinned by the
reference's own vectors where they exist and by independent implementations elsewhere
(tests/test_oracle_independent.py); with the file present, oracle AND HIP path are compared
with the packages' own outputs stage by stage.

The calls mirror the reference's call sites: core.py:48-58 (wpe_v8), core.py:154-214
(CACGMMTrainer.fit / predict), core.py:305-321 (stft / istft),
speech_enhancement/beamforming_wrapper.py:49-100 (PSD, MVDR-Souden, BAN, GEV, apply).
Inputs are seeded and small; they are stored next to the outputs.
"""
import sys
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parents[2]
OUT = REPO / 'tests' / 'golden' / 'upstream_stages.npz'


def crandn(rng, *shape):
    return rng.standard_normal(shape) + 1j * rng.standard_normal(shape)


def main():
    try:
        import nara_wpe
        import nara_wpe.utils
        import nara_wpe.wpe
        from pb_bss.distribution import CACGMMTrainer
        from pb_bss.extraction import beamformer
    except ImportError as e:
        sys.exit(f'upstream packages not importable ({e}); nothing written')

    rng = np.random.default_rng(20260928)
    out = {'versions': np.array(str(getattr(nara_wpe, '__version__', 'unknown')))}

    # ---- STFT / iSTFT (core.py:305-321): size 1024, shift 256, fading=True
    x = rng.standard_normal((3, 9000))
    X = nara_wpe.utils.stft(x, size=1024, shift=256, fading=True)
    out['stft/x'] = x
    out['stft/X'] = X
    Xmod = X[0] * (0.5 + rng.uniform(size=X[0].shape))            # an inconsistent STFT
    out['istft/X'] = Xmod
    out['istft/x'] = nara_wpe.utils.istft(Xmod, size=1024, shift=256, fading=True)

    # ---- WPE (core.py:48-58): (F, D, T) in, taps 4, delay 2, 3 iterations; psd_context 0 and 2
    F, D, T = 4, 5, 260
    S = crandn(rng, F, 1, T + 6)
    h = crandn(rng, F, D, 6) * np.exp(-np.arange(6))
    Y = sum(h[..., k:k + 1] * S[..., 6 - k:6 - k + T] for k in range(6)) + 0.05 * crandn(rng, F, D, T)
    out['wpe/Y'] = Y
    for ctx in (0, 2):
        out[f'wpe/X_ctx{ctx}'] = nara_wpe.wpe.wpe_v8(Y, taps=4, delay=2, iterations=3,
                                                     psd_context=ctx)

    # ---- guided CACGMM (core.py:154-214), one frequency at a time like GSS.__call__
    K, Tg = 3, 140
    act = np.zeros((K, Tg), bool)
    act[0, 10:80] = True
    act[1, 60:130] = True
    act[2] = True
    Obs = 0.3 * crandn(rng, F, Tg, D)                              # Obs.T[f] = (T, D)
    for k in range(2):
        Obs += crandn(rng, F, 1, D) * crandn(rng, F, Tg, 1) * act[k][None, :, None]
    init = np.where(act, 1.0, 1e-10)
    init = init / init.sum(axis=0, keepdims=True)
    out['em/Obs'] = Obs
    out['em/activity'] = act
    for iterations, post in ((5, 1), (4, 0), (3, 3)):
        aff = []
        for f in range(F):
            tr = CACGMMTrainer()
            cur = tr.fit(y=Obs[f], initialization=init, iterations=iterations,
                         source_activity_mask=act)
            if post != 0:
                if post != 1:
                    cur = tr.fit(y=Obs[f], initialization=cur, iterations=post - 1)
                aff.append(cur.predict(Obs[f]))
            else:
                aff.append(cur.predict(Obs[f], source_activity_mask=act))
        out[f'em/posterior_{iterations}_{post}'] = np.array(aff)  # (F, K, T)

    # ---- beamformers (beamforming_wrapper.py:49-100): Y (F, D, T), masks (F, T)
    Yb = crandn(rng, F, D, Tg) * 0.5 + crandn(rng, F, D, 1) * crandn(rng, F, 1, Tg)
    mx = rng.uniform(size=(F, Tg))
    mn = 1 - mx
    cov_x = beamformer.get_power_spectral_density_matrix(Yb, mx)
    cov_n = beamformer.get_power_spectral_density_matrix(Yb, mn)
    w = beamformer.get_mvdr_vector_souden(cov_x, cov_n, eps=1e-10)
    w_ban = beamformer.blind_analytic_normalization(w, cov_n)
    out.update({'bf/Y': Yb, 'bf/mx': mx, 'bf/mn': mn, 'bf/cov_x': cov_x, 'bf/cov_n': cov_n,
                'bf/w_mvdr': w, 'bf/w_mvdr_ban': w_ban,
                'bf/X_mvdr_ban': beamformer.apply_beamforming_vector(w_ban, Yb).T})
    try:
        w_gev = beamformer.get_gev_vector(cov_x, cov_n, force_cython=True)
    except Exception:                                             # no compiled extension
        w_gev = beamformer.get_gev_vector(cov_x, cov_n)
    w_gev_ban = beamformer.blind_analytic_normalization(w_gev, cov_n)
    out.update({'bf/w_gev': w_gev,
                'bf/X_gev_ban': beamformer.apply_beamforming_vector(w_gev_ban, Yb).T})

    np.savez_compressed(OUT, **out)
    print('wrote', OUT)


if __name__ == '__main__':
    main()
