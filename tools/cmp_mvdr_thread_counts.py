"""mvdr_solve_kernel with 256 threads against a one-wave build: the same bits?

    bash tools/build_variant.sh mvdr_nt64 -DGSS_MVDR_NT=64
    python tools/cmp_mvdr_thread_counts.py          (on a GPU box)

Runs MVDR-Souden + BAN on random inputs with 4 - 29 channels and on an input with a dead channel
(the pseudo-inverse path) under both libraries (GSS_HIP_LIBRARY) and compares outputs and
reference channels byte for byte."""
import os, sys, subprocess, json
sys.path.insert(0, '.'); sys.path.insert(0, 'oracle')
import numpy as np
if len(sys.argv) > 1:
    from pb_chime5_amd import ops
    rng = np.random.default_rng(5)
    out = {}
    for (D, T, F) in ((24, 300, 40), (20, 200, 16), (12, 150, 16), (4, 80, 9), (7, 129, 6), (29, 200, 3)):
        Y = rng.standard_normal((D, T, F)) + 1j * rng.standard_normal((D, T, F))
        Y += (rng.standard_normal((D, 1, F)) + 1j * rng.standard_normal((D, 1, F))) * (rng.standard_normal((1, T, F)) + 1j * rng.standard_normal((1, T, F))) * 3
        tm = rng.uniform(size=(T, F)); dm = rng.uniform(size=(T, F)) * (1 - tm)
        X, ref = ops.mvdr_souden_from_masks(Y, tm, dm, ban=True, return_ref_channel=True)
        out[f'X{D}'] = X; out[f'r{D}'] = np.array(ref)
    # a singular Phi_N (dead channel): the pseudo-inverse path
    Y = rng.standard_normal((6, 90, 5)) + 1j * rng.standard_normal((6, 90, 5)); Y[2] = 0
    tm = rng.uniform(size=(90, 5)); dm = 1 - tm
    out['Xs'] = ops.mvdr_souden_from_masks(Y, tm, dm, ban=True)
    np.savez(sys.argv[1], **out)
else:
    a = '/tmp/mvdr_a.npz'; b = '/tmp/mvdr_b.npz'
    subprocess.run([sys.executable, os.path.abspath(__file__), a], check=True)
    subprocess.run([sys.executable, os.path.abspath(__file__), b], check=True, env=dict(os.environ, GSS_HIP_LIBRARY='pb_chime5_amd/lib/variants/libgss_mvdr_nt64.so'))
    A, B = np.load(a), np.load(b)
    for k in A.files:
        same = A[k].tobytes() == B[k].tobytes()
        print(k, 'bit-identical' if same else 'DIFFERENT max %.3e' % np.nanmax(np.abs(A[k] - B[k])), 'nan' if np.isnan(A[k]).any() else '')
