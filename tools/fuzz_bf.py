"""Random beamformer-stage inputs against the oracle: 1 - 29 channels, frame counts from fewer
than channels (rank-deficient PSD matrices) to a few hundred, masks that are uniform, sparse,
tiny or zero in whole frequencies; MVDR-Souden (+/- BAN) and GEV (+/- BAN).  A case counts as
failed when
* the two disagree where the noise PSD matrix is well conditioned (cond < 1e6: 1e-6);
* in frequencies where it is not (the reference's float64 evaluation of its own formulas is
  then decided by rounding), the HIP result is further than max(1e-6, 4 cond eps) from the
  80-bit evaluation of the reference's formulas (tests/ext_precision.py; MVDR +/- BAN, GEV + BAN);
* only one side raises -- unless the ORACLE'S OWN decision flips when its input is perturbed
  in the last bit (three trials), which is counted as "one-sided, certified rounding".
python tools/fuzz_bf.py [SEED] [CASES]"""
import sys
import warnings
from pathlib import Path

import numpy as np

R = Path(__file__).resolve().parents[1]
for p in (str(R), str(R / 'oracle'), str(R / 'tests')):
    sys.path.insert(0, p)


def main():
    import gss_oracle as oracle
    import ext_precision as ext
    from pb_chime5_amd import ops
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    cases = int(sys.argv[2]) if len(sys.argv) > 2 else 150
    rng = np.random.default_rng(seed)
    bad = 0
    notes = {}
    warnings.simplefilter('ignore')
    for case in range(cases):
        D = int(rng.integers(1, 30)); F = int(rng.integers(1, 12))
        T = int(rng.integers(max(2, D // 2), 4 * D + 40))
        K = int(rng.integers(1, 4))
        # K point sources + sensor noise
        A = rng.standard_normal((F, D, K)) + 1j * rng.standard_normal((F, D, K))
        S = rng.standard_normal((F, K, T)) + 1j * rng.standard_normal((F, K, T))
        noise = 10.0 ** rng.uniform(-3, 0)
        Y = A @ S + noise * (rng.standard_normal((F, D, T)) + 1j * rng.standard_normal((F, D, T)))
        Y = np.ascontiguousarray(Y.transpose(1, 2, 0))                         # (D, T, F)
        kind = int(rng.integers(0, 4))
        xm = rng.uniform(size=(T, F))
        if kind == 1:
            xm *= rng.uniform(size=(T, F)) < 0.2                               # sparse
        elif kind == 2:
            xm *= 10.0 ** rng.uniform(-12, -3)                                  # tiny
        nm = 1 - xm if kind != 2 else rng.uniform(size=(T, F))
        if kind == 3 and F > 1:
            nm[:, int(rng.integers(0, F))] = 0.0                                # a dead frequency
        ban = bool(rng.integers(0, 2))
        bf = ['mvdr', 'gev'][int(rng.integers(0, 2))]
        tag = dict(case=case, D=D, T=T, F=F, kind=kind, ban=ban, bf=bf, noise=round(noise, 4))
        res = {}
        for side in ('oracle', 'gpu'):
            try:
                if bf == 'mvdr':
                    if side == 'oracle':
                        X, det = oracle.beamform_mvdr_souden_from_masks(Y, xm, nm, ban=ban, return_details=True)
                        res[side] = (X, det['ref_channel'])
                    else:
                        res[side] = ops.mvdr_souden_from_masks(Y, xm, nm, ban=ban, return_ref_channel=True)
                else:
                    fn = oracle.beamform_gev_from_masks if side == 'oracle' else ops.gev_from_masks
                    res[side] = (fn(Y, xm, nm, ban=ban), 0)
            except (AssertionError, np.linalg.LinAlgError) as e:
                res[side] = type(e).__name__
        o, g = res['oracle'], res['gpu']
        if isinstance(o, str) or isinstance(g, str):
            key = f'raises: oracle {o if isinstance(o, str) else "-"}, gpu {g if isinstance(g, str) else "-"}'
            notes[key] = notes.get(key, 0) + 1
            if isinstance(o, str) != isinstance(g, str):
                # does the oracle's own decision (raise / return) survive a last-bit perturbation?
                flips = 0
                for trial in range(3):
                    Yp = Y * (1.0 + 2.2e-16 * np.random.default_rng(1000 * seed + 10 * case + trial)
                              .standard_normal(Y.shape))
                    try:
                        if bf == 'mvdr':
                            oracle.beamform_mvdr_souden_from_masks(Yp, xm, nm, ban=ban)
                        else:
                            oracle.beamform_gev_from_masks(Yp, xm, nm, ban=ban)
                        raised = False
                    except (AssertionError, np.linalg.LinAlgError):
                        raised = True
                    flips += raised != isinstance(o, str)
                if flips:
                    notes['one-sided, certified rounding'] = notes.get('one-sided, certified rounding', 0) + 1
                else:
                    print('only one side raises and the oracle\'s decision is stable', key, tag)
                    bad += 1
            continue
        cn = oracle.get_power_spectral_density_matrix(Y.transpose(2, 0, 1), nm.T)
        cx = oracle.get_power_spectral_density_matrix(Y.transpose(2, 0, 1), xm.T)
        cond = np.linalg.cond(cn)
        good = cond < 1e6           # BAN in the reference loses cond^2 eps
        if bf == 'gev' and D > 1:
            lam = np.sort(np.linalg.eigvals(np.linalg.solve(cn[good], cx[good])).real, axis=-1)
            sep = np.zeros(F, bool)
            sep[good] = lam[:, -1] - lam[:, -2] > 1e-6 * np.abs(lam[:, -1])
            good &= sep
        # frequencies the literal oracle cannot decide: the extended-precision referee
        hard = ~good & np.isfinite(cond) & (cond < 1e13) & ~np.isnan(g[0]).any(axis=0)
        if bf == 'gev' and not ban:
            hard[:] = False                        # (the referee has GEV + BAN only)
        ref_bad = False
        for f in np.flatnonzero(hard):
            if bf == 'mvdr':
                cov_x, cov_n = ext.psd(Y[..., f], xm[:, f]), ext.psd(Y[..., f], nm[:, f])
                w = ext.souden_matrix(cov_x, cov_n)[:, g[1]]
                if ban:
                    w = ext.ban(w, cov_n)
                want = (w.conj() @ Y[..., f].astype(ext.CLD)).astype(np.complex128)
            else:
                want = ext.gev_ban_output(Y[..., f], xm[:, f], nm[:, f])
            if not np.all(np.isfinite(want)) or np.linalg.norm(want) == 0:
                continue
            e = np.linalg.norm(np.abs(g[0][:, f]) - np.abs(want)) / np.linalg.norm(want)
            if not e < max(1e-6, 4 * cond[f] * 2.2e-16):
                print('referee', e, 'cond', cond[f], 'bin', f, tag)
                ref_bad = True
            else:
                notes['bins held to the referee'] = notes.get('bins held to the referee', 0) + 1
        if ref_bad:
            bad += 1
            continue
        if not good.any():
            key = 'held to the referee only' if hard.any() else 'nothing comparable'
            notes[key] = notes.get(key, 0) + 1
            continue
        if bf == 'mvdr' and o[1] != g[1]:
            notes['reference channel differs'] = notes.get('reference channel differs', 0) + 1
            if good.all() and cond.max() < 1e4:
                print('reference channel differs on a well-conditioned input', o[1], g[1], tag)
                bad += 1
            continue
        # a frequency whose target mask is all zero: Phi_X = 0, w = 0 and BAN's 0 / 0 is NaN in
        # the reference -- NaN in the same frequencies here
        nan_o, nan_g = np.isnan(o[0]).any(axis=0), np.isnan(g[0]).any(axis=0)
        if not np.array_equal(nan_o, nan_g):
            print('NaN in different frequencies', np.flatnonzero(nan_o), np.flatnonzero(nan_g), tag)
            bad += 1
            continue
        good &= ~nan_o
        if not good.any():
            notes['nothing comparable'] = notes.get('nothing comparable', 0) + 1
            continue
        n = np.linalg.norm
        err = n(np.abs(g[0][:, good]) - np.abs(o[0][:, good])) / max(n(o[0][:, good]), 1e-300)
        if not err < 1e-6:
            print('output', err, 'cond max over compared', cond[good].max(), tag)
            bad += 1
        else:
            notes['ok'] = notes.get('ok', 0) + 1
    print('beamformer fuzz: seed', seed, 'cases', cases, 'failures', bad, notes)


if __name__ == '__main__':
    main()
