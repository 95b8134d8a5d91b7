"""oracle.stable_solve against the REFERENCE'S own pb_chime5.math.solve.stable_solve
(math/solve.py:7-114, build container only) on random batched systems: well conditioned,
exactly singular (zero rows / columns, repeated columns: the lstsq branch) and mixed batches.
    python tests/golden/fuzz_solve_vs_reference.py [SEED] [CASES]"""
import sys
import importlib.util
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent.parent / 'oracle'))


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    cases = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    rng = np.random.default_rng(seed)
    spec = importlib.util.spec_from_file_location('ref_solve', '/root/reference/pb_chime5/math/solve.py')
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    import gss_oracle as oracle
    bad = 0
    for case in range(cases):
        lead = tuple(int(x) for x in rng.integers(1, 4, size=int(rng.integers(0, 3))))
        n = int(rng.integers(1, 9)); m = int(rng.integers(1, 5))
        A = rng.standard_normal(lead + (n, n)) + 1j * rng.standard_normal(lead + (n, n))
        B = rng.standard_normal(lead + (n, m)) + 1j * rng.standard_normal(lead + (n, m))
        kind = int(rng.integers(0, 4))
        if kind == 1 and n > 1:          # one matrix of the batch exactly singular
            idx = tuple(int(rng.integers(0, s)) for s in lead)
            A[idx + (slice(None), 0)] = 0
            A[idx + (0, slice(None))] = 0
        elif kind == 2 and n > 1:        # repeated column everywhere
            A[..., :, 1] = A[..., :, 0]
        elif kind == 3:
            A = A @ np.conj(np.swapaxes(A, -1, -2))       # Hermitian positive definite
        res = {}
        for side, fn in (('reference', ref.stable_solve), ('oracle', oracle.stable_solve)):
            try:
                res[side] = np.asarray(fn(A, B))
            except Exception as e:
                res[side] = type(e).__name__
        r, o = res['reference'], res['oracle']
        tag = dict(case=case, lead=lead, n=n, m=m, kind=kind)
        if isinstance(r, str) or isinstance(o, str):
            if r != o if (isinstance(r, str) and isinstance(o, str)) else True:
                print('exceptions differ', r if isinstance(r, str) else 'ok', o if isinstance(o, str) else 'ok', tag)
                bad += 1
            continue
        if r.shape != o.shape or not np.array_equal(r, o, equal_nan=True):
            err = np.max(np.abs(r - o)) / max(np.max(np.abs(r)), 1e-300) if r.shape == o.shape else None
            if err is None or err > 1e-12:
                print('differs', err, tag)
                bad += 1
    print('stable_solve fuzz: seed', seed, 'cases', cases, 'failures', bad)


if __name__ == '__main__':
    main()
