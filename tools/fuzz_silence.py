"""Digital silence through the whole pipeline against the oracle: the wide parameter draws of
tests/fuzz_params.py (channel / class / frame counts, window lengths, WPE and EM settings, MVDR /
GEV / postfilter), every case with a dropped, zero-filled block in the recording -- every
channel for a while, the channels of one "array", everything before or after some point, two
blocks -- and judged by the pipeline sweep's own rules (tests/test_gpu_pipeline.py::_fuzz_case:
well-conditioned bins against the literal oracle, the others against the extended-precision
referees, one-sided exceptions, reference-channel ties).  Exact zeros are where reformulations
stop being equivalent (the clamp of the CACG quadratic form: EXPERIMENTS round 6, item 12), and
no other sweep draws them.
    python tools/fuzz_silence.py [SEED] [CASES]"""
import os
import sys
import warnings
from pathlib import Path

import numpy as np

R = Path(__file__).resolve().parents[1]
for p in (str(R), str(R / 'oracle'), str(R / 'tests')):
    sys.path.insert(0, p)


def replay_wpe(tp, ctx, case, D, K, N, ctx_s, kw, mutate):
    import ext_precision
    import gss_oracle as oracle
    from pb_chime5_amd import ops, synthetic
    u = synthetic.tiny(seed=5000 + case, num_channels=D, num_samples=N, num_speakers=K - 1,
                       context=ctx_s, noise=5e-2)
    mutate(u)
    Y = oracle.stft(u.obs, kw.get('stft_size', 1024), kw.get('stft_shift', 256),
                    fading=kw.get('stft_fading', True))
    Xo = oracle.wpe_block(Y, kw['wpe_taps'], kw['wpe_delay'], kw['wpe_iterations'],
                          kw.get('wpe_psd_context', 0))
    Xg = ops.wpe_dtf(Y, kw['wpe_taps'], kw['wpe_delay'], kw['wpe_iterations'],
                     kw.get('wpe_psd_context', 0), ctx=ctx)
    n = np.linalg.norm
    per = np.array([n(Xg[..., f] - Xo[..., f]) / max(n(Xo[..., f]), 1e-300) for f in range(Xg.shape[-1])])
    print('replay: WPE GPU - oracle', tp.rel_err(Xg, Xo), 'GSS_VARIANT', os.environ.get('GSS_VARIANT', ''))
    for f in np.argsort(per)[::-1][:4]:
        Yf = np.ascontiguousarray(Y[..., f])
        Yt = oracle.build_y_tilde(Yf, kw['wpe_taps'], kw['wpe_delay'])
        Xt = ext_precision.wpe(Yf, Yt, kw['wpe_iterations'])[-1]
        w = oracle.get_power_inverse(Yf)
        R = (Yt * w) @ Yt.conj().T
        print('  frequency', int(f), 'cond(R)', f'{np.linalg.cond(R):.1e}', 'GPU - referee',
              f'{n(Xg[..., f] - Xt) / n(Xt):.2e}', 'oracle - referee', f'{n(Xo[..., f] - Xt) / n(Xt):.2e}')


def main():
    import fuzz_params
    import test_gpu_pipeline as tp
    from pb_chime5_amd._capi import default_context
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    cases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    warnings.simplefilter('ignore')
    ctx = default_context(0)
    mismatches, failures, notes, done, skipped = [], 0, {}, 0, 0
    for case, D, K, N, ctx_s, kw in fuzz_params.fuzz_cases(seed, cases, True):
        zr = np.random.default_rng([seed, case, 13])
        kind = int(zr.integers(0, 5))
        a = int(zr.integers(0, max(1, N - 6000)))
        b = a + int(zr.integers(1500, 12000))
        a2 = int(zr.integers(0, max(1, N - 3000)))
        # what is left of the recording must still carry the case: a WPE that stays
        # overdetermined on the frames that are not silent (else it is the documented
        # "another minimiser" case of INTEGRATION.md section 2, not this sweep's subject), a PSD
        # matrix of full rank -- and not the all-zero recording (NaN on both sides, its own test)
        silent = {0: min(b, N) - a, 1: min(b, N), 2: N - a, 3: 0, 4: min(b, N) - a + 2500}[kind]
        shift = kw.get('stft_shift', 256)
        frames_left = (N - silent) // shift
        need = 3 * kw['wpe_taps'] * D + 10 if kw['wpe'] else D + 10
        if frames_left < need:
            skipped += 1
            continue

        def mutate(u, kind=kind, a=a, b=b, a2=a2, D=D):
            if kind == 0:
                u.obs[:, a:b] = 0.0                               # every channel for a while
            elif kind == 1:
                u.obs[:, :b] = 0.0                                # the recording starts late
            elif kind == 2:
                u.obs[:, a:] = 0.0                                # ... or ends early
            elif kind == 3:
                u.obs[:max(1, D // 2), a:b] = 0.0                 # one "array" drops out
            else:
                u.obs[:, a:b] = 0.0                               # two blocks
                u.obs[:, a2:a2 + 2500] = 0.0
        if os.environ.get('GSS_FUZZ_ONLY'):
            # replay one case: the WPE stage against the 80-bit iteration, frequency by frequency
            if case != int(os.environ['GSS_FUZZ_ONLY']):
                continue
            replay_wpe(tp, ctx, case, D, K, N, ctx_s, kw, mutate)
        try:
            note = tp._fuzz_case(ctx, mismatches, case, D, K, N, ctx_s, kw, wide=True, mutate=mutate)
        except AssertionError as e:
            print('FAILED', dict(case=case, kind=kind, a=a, b=b), str(e)[:700])
            failures += 1
            note = 'failed'
        note = f'{note}'
        notes[note] = notes.get(note, 0) + 1
        done += 1
    print('silence fuzz: seed', seed, 'cases', done, 'failures', failures, notes,
          'skipped (too little left of the recording)', skipped)
    print('reference-channel mismatches:', mismatches)


if __name__ == '__main__':
    main()
