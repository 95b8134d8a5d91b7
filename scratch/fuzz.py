"""Random shapes through the fused pipeline vs the oracle."""
import sys, time
sys.path.insert(0, 'oracle'); sys.path.insert(0, '.')
import numpy as np
import gss_oracle as oracle
from pb_chime5_amd import ops, synthetic
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 30
bad = 0
for case in range(n_cases):
    D = int(rng.integers(2, 30)); K = int(rng.integers(2, 7))
    N = int(rng.integers(9000, 40000)); ctx_s = int(rng.integers(0, 3000))
    taps = int(rng.integers(1, 5)); delay = int(rng.integers(1, 4)); wit = int(rng.integers(1, 3))
    bss = int(rng.integers(1, 5)); post = int(rng.integers(0, 3))
    bf = ['mvdrSouden_ban', 'gev_ban', 'ch2', 'sum'][int(rng.integers(0, 4))]
    if bf == 'ch2' and D < 3: bf = 'sum'
    pf = [None, 'mask_mul'][int(rng.integers(0, 2))]
    wpe = bool(rng.integers(0, 4) > 0)
    u = synthetic.tiny(seed=1000 + case, num_channels=D, num_samples=N, num_speakers=K - 1,
                       context=ctx_s, noise=5e-2)
    kw = dict(wpe=wpe, wpe_taps=taps, wpe_delay=delay, wpe_iterations=wit, bss_iterations=bss,
              bss_iterations_post=post, bf=bf, postfilter=pf)
    T = oracle.stft(u.obs[:1]).shape[1]
    tag = f'case {case:2d} D={D:2d} K={K} N={N:5d} T={T:3d} ctx={ctx_s:4d} wpe={int(wpe)} taps={taps} delay={delay} it={wit} bss={bss} post={post} bf={bf:14s} pf={pf}'
    if wpe and T < 3 * taps * D + 10:
        print(tag, 'SKIP (too few frames for WPE)'); continue
    try:
        got, det = ops.enhance_observation(u.obs, u.activity_array, u.target_index, ctx_s, ctx_s, debug=True, **kw)
        want, wdet = oracle.enhance_observation(u.obs, u.activity_array, u.target_index, u.ex, return_details=True,
                                                gss_fn=oracle.gss_block_batched, **kw)
    except Exception as e:
        print(tag, 'EXC', type(e).__name__, str(e)[:80]); bad += 1; continue
    e_obs = np.linalg.norm(det['Obs'] - wdet['Obs']) / np.linalg.norm(wdet['Obs'])
    ok_ref = bf not in ('mvdrSouden_ban',) or det['ref_channel'] == wdet['ref_channel']
    if bf == 'gev_ban':      # the principal eigenvector has an arbitrary phase per frequency
        got, want = np.abs(det['X_hat']), np.abs(wdet['X_hat'])
    e_x = np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-300) if np.isfinite(want).all() else float('nan')
    status = 'ok' if (np.isfinite(got).all() == np.isfinite(want).all() and (not np.isfinite(e_x) or e_x < 1e-4) and ok_ref) else 'MISMATCH'
    if status != 'ok': bad += 1
    print(tag, f'| Obs {e_obs:.1e} x {e_x:.1e} ref {"=" if ok_ref else "!="} {status}')
print('mismatches:', bad)
