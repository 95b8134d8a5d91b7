"""One frequency of a fuzz case through the EM in several configurations (GPU, forced
eigendecomposition, oracle, brute force), to localise a disagreement.
    GSS_FUZZ_SEED=202 GSS_FUZZ_WIDE=1 python tools/em_bin_probe.py CASE BIN"""
import os
import subprocess
import sys
from pathlib import Path

import numpy as np

R = Path(__file__).resolve().parents[1]
for p in (str(R), str(R / 'oracle'), str(R / 'tests')):
    sys.path.insert(0, p)


def main():
    import gss_oracle as oracle
    from pb_chime5_amd import ops, synthetic
    from test_oracle_independent import brute_force_guided_em
    if len(sys.argv) > 3:            # child: GPU posteriors of a saved bin with the given settings
        d = np.load(sys.argv[3])
        g = ops.cacgmm_posteriors(d['Of'], d['act'], int(sys.argv[4]), int(sys.argv[5]))[..., 0]
        np.save(sys.argv[6], g)
        return
    import fuzz_params
    want_case, f = int(sys.argv[1]), int(sys.argv[2])
    seed, _, wide = fuzz_params.from_environment()
    case, D, K, N, ctx_s, kw = [c for c in fuzz_params.fuzz_cases(seed, want_case + 1, wide)
                                if c[0] == want_case][0]
    size, shift = kw.get('stft_size', 1024), kw.get('stft_shift', 256)
    fading = kw.get('stft_fading', True)
    u = synthetic.tiny(seed=5000 + case, num_channels=D, num_samples=N, num_speakers=K - 1,
                       context=ctx_s, noise=5e-2)
    Y = oracle.stft(u.obs, size, shift, fading=fading)
    act = oracle.activity_time_to_frequency(np.asarray(u.activity_array), size, shift, fading,
                                            stft_pad=True)[:, :Y.shape[1]]
    Of = np.ascontiguousarray(Y[..., f:f + 1])
    out = R / 'gpurun_out'
    out.mkdir(exist_ok=True)
    np.savez(out / f'em_bin_{want_case}_{f}.npz', Of=Of, act=act)
    print('D', D, 'K', K, 'T', Y.shape[1], 'active frames per class', act.sum(axis=1))
    for it, post in ((1, 0), (1, 1), (2, 0), (2, 1), (2, 2), (3, 1), (4, 2)):
        o = oracle.gss_block_batched(Of, act, iterations=it, iterations_post=post)[..., 0]
        b = brute_force_guided_em(np.ascontiguousarray(Of[..., 0].T), act, it, post)
        row = [f'iterations {it} post {post}: oracle-brute {np.max(np.abs(o - b)):.1e}']
        for tag, env in (('GPU', {}), ('GPU eigh', {'GSS_VARIANT': 'force_eigh'}),
                         ('GPU LDS E-step', {'GSS_VARIANT': 'estep_lds'})):
            tmp = f'/tmp/em_probe_{tag.replace(" ", "_")}.npy'
            subprocess.run([sys.executable, __file__, '0', '0', str(out / f'em_bin_{want_case}_{f}.npz'),
                            str(it), str(post), tmp], env={**os.environ, **env}, check=True)
            g = np.load(tmp)
            row.append(f'{tag}-oracle {np.max(np.abs(g - o)):.1e}')
        print('  '.join(row))


if __name__ == '__main__':
    main()
