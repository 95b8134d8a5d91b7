"""Replay one case of tests/test_gpu_pipeline.py::test_random_shapes_against_oracle (same
generator, GSS_FUZZ_SEED / GSS_FUZZ_WIDE as there) stage by stage, with the extended-precision
referees of tests/ext_precision.py where GPU and oracle disagree:

    GSS_FUZZ_SEED=303 GSS_FUZZ_WIDE=1 python tools/fuzz_case.py CASE"""
import os
import sys
from pathlib import Path

import numpy as np

R = Path(__file__).resolve().parents[1]
for p in (str(R), str(R / 'oracle'), str(R / 'tests')):
    sys.path.insert(0, p)


def main():
    import ext_precision
    import gss_oracle as oracle
    from pb_chime5_amd import ops, synthetic
    import fuzz_params
    want_case = int(sys.argv[1])
    seed, _, wide = fuzz_params.from_environment()
    hit = [c for c in fuzz_params.fuzz_cases(seed, want_case + 1, wide) if c[0] == want_case]
    assert hit, f'case {want_case} is not a runnable draw of seed {seed}'
    case, D, K, N, ctx_s, kw = hit[0]
    taps, delay, wit = kw['wpe_taps'], kw['wpe_delay'], kw['wpe_iterations']
    bss, post, bf, pf, wpe = (kw['bss_iterations'], kw['bss_iterations_post'], kw['bf'],
                              kw['postfilter'], kw['wpe'])
    size, shift = kw.get('stft_size', 1024), kw.get('stft_shift', 256)
    fading, psd_context = kw.get('stft_fading', True), kw.get('wpe_psd_context', 0)
    print(dict(case=case, D=D, K=K, N=N, ctx=ctx_s), kw)
    u = synthetic.tiny(seed=5000 + case, num_channels=D, num_samples=N, num_speakers=K - 1,
                       context=ctx_s, noise=5e-2)
    got, det = ops.enhance_observation(u.obs, u.activity_array, u.target_index, ctx_s, ctx_s,
                                       debug=True, **kw)
    want, wdet = oracle.enhance_observation(u.obs, u.activity_array, u.target_index, u.ex,
                                            return_details=True, gss_fn=oracle.gss_block_batched,
                                            **kw)
    n = np.linalg.norm
    Xg, Xo = det['Obs'], wdet['Obs']
    F = Xg.shape[-1]
    print('frames', Xg.shape[1], ' WPE output, GPU vs oracle:', n(Xg - Xo) / n(Xo))
    if wpe:
        Y = oracle.stft(u.obs, size, shift, fading=fading)
        per = np.array([n(Xg[..., f] - Xo[..., f]) / n(Xo[..., f]) for f in range(F)])
        for f in np.argsort(per)[::-1][:4]:
            Yf = np.ascontiguousarray(Y[..., f])
            Yt = oracle.build_y_tilde(Yf, taps, delay)
            Xt = ext_precision.wpe(Yf, Yt, wit)[-1]          # (psd_context = 0 only)
            cond = np.linalg.cond((Yt * oracle.get_power_inverse(Yf)) @ Yt.conj().T)
            print(f'  bin {f}: GPU-oracle {per[f]:.2e}  GPU-truth {n(Xg[..., f] - Xt) / n(Xt):.2e}  '
                  f'oracle-truth {n(Xo[..., f] - Xt) / n(Xt):.2e}  cond(R) {cond:.2e}')
    tm, dm = det['target_mask'], det['distortion_mask']
    dmask = np.max(np.abs(tm - wdet['target_mask']), axis=0)
    print('target mask, GPU vs oracle: max abs diff %.2e (bin %d)' % (dmask.max(), int(np.argmax(dmask))))
    if dmask.max() > 1e-6:
        # the EM of the worst frequency three ways: GPU, oracle, and the brute-force guided EM of
        # tests/test_oracle_independent.py (per-frame inverses and log-determinants, float64)
        from test_oracle_independent import brute_force_guided_em
        f = int(np.argmax(dmask))
        act = wdet['activity_freq'][:, :Xo.shape[1]]
        Of = np.ascontiguousarray(Xo[..., f:f + 1])
        g = ops.cacgmm_posteriors(Of, act, bss, post)[..., 0]
        o = oracle.gss_block_batched(Of, act, iterations=bss, iterations_post=post)[..., 0]
        b = brute_force_guided_em(np.ascontiguousarray(Xo[..., f].T), act, bss, post)
        print(f'  EM of bin {f} alone: GPU-oracle {np.max(np.abs(g - o)):.2e}  GPU-brute force '
              f'{np.max(np.abs(g - b)):.2e}  oracle-brute force {np.max(np.abs(o - b)):.2e}')
    cov_n = oracle.get_power_spectral_density_matrix(Xo.transpose(2, 0, 1), dm.T)
    cov_x = oracle.get_power_spectral_density_matrix(Xo.transpose(2, 0, 1), tm.T)
    cond = np.linalg.cond(cov_n)
    print('cond(Phi_N): median %.2e max %.2e; bins < 1e8: %d of %d' %
          (np.median(cond), cond.max(), int((cond < 1e8).sum()), F))
    per = np.array([n(np.abs(det['X_hat'][:, f]) - np.abs(wdet['X_hat'][:, f])) /
                    max(n(wdet['X_hat'][:, f]), 1e-300) for f in range(F)])
    print('|X_hat| GPU vs oracle: global', n(np.abs(det['X_hat']) - np.abs(wdet['X_hat'])) /
          n(wdet['X_hat']), ' worst bins', np.argsort(per)[::-1][:6], np.sort(per)[::-1][:6])
    if bf in ('gev_ban', 'mvdrSouden_ban'):
        for f in np.argsort(per)[::-1][:5]:
            if bf == 'gev_ban':
                ref = ext_precision.gev_ban_output(Xg[..., f], tm[:, f], dm[:, f])
                lam = np.sort(np.linalg.eigvals(np.linalg.solve(cov_n[f], cov_x[f])).real)[::-1]
                extra = 'two largest generalised eigenvalues %.6e %.6e' % (lam[0], lam[1])
            else:
                ref = ext_precision.mvdr_souden_ban_output(Xg[..., f], tm[:, f], dm[:, f],
                                                           det['ref_channel'])
                extra = ''
            if pf == 'mask_mul':
                ref = ref * tm[:, f]
            eg = n(np.abs(det['X_hat'][:, f]) - np.abs(ref)) / n(ref)
            eo = n(np.abs(wdet['X_hat'][:, f]) - np.abs(ref)) / n(ref)
            print(f'  bin {f}: cond {cond[f]:.2e}  mask diff {dmask[f]:.1e}  GPU-referee {eg:.2e}  oracle-referee {eo:.2e}  {extra}')


if __name__ == '__main__':
    main()
