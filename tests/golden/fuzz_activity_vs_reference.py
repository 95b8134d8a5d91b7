"""oracle.activity_time_to_frequency against the REFERENCE'S own function
(database/chime5/database.py:409-463, importable in the build container only) on random activity
tracks, window lengths, shifts, fading and padding: the oracle restates it, this checks the
restatement beyond the doctest vectors.  The GPU kernel is held bit-exact to the oracle by
tests/test_gpu_stages.py::test_activity_bit_exact_random.
    python tests/golden/fuzz_activity_vs_reference.py [SEED] [CASES]"""
import sys
import tempfile
import types
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(HERE.parent))
import make_golden as mg  # noqa: E402
import make_golden_session as mgs  # noqa: E402


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    cases = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    rng = np.random.default_rng(seed)
    with tempfile.TemporaryDirectory() as tmp:
        tmp = Path(tmp)
        ref = mg._prepare_reference(tmp)
        mg._register_stubs()
        mg._module('lazy_dataset', from_dict=lambda d: mgs._Dataset(d.values()),
                   concatenate=mgs._concatenate)
        mg._module('soundfile', SoundFile=mgs._SoundFile)
        sys.path.insert(0, str(ref))
        pkg = types.ModuleType('pb_chime5')
        pkg.__path__ = [str(ref / 'pb_chime5')]
        pkg.git_root = ref
        sys.modules['pb_chime5'] = pkg
        import gss_oracle as oracle
        from pb_chime5.database.chime5.database import activity_time_to_frequency as ref_fn
        bad = 0
        for case in range(cases):
            size = int(rng.choice([4, 16, 64, 400, 512, 1024, 2048]))
            shift = size // int(rng.choice([d for d in (1, 2, 4, 8) if size % d == 0]))
            fading = bool(rng.integers(0, 2))
            pad = bool(rng.integers(0, 4) > 0)
            N = int(rng.choice([size, size + 1, 2 * size - 1, int(rng.integers(size, 20 * size + 40))]))
            K = int(rng.integers(1, 4))
            act = rng.uniform(size=(K, N)) < rng.uniform(0.0, 0.3)
            # runs rather than salt and pepper
            act = np.array([np.convolve(a, np.ones(int(rng.integers(1, size))), 'same') > 0 for a in act])
            if rng.integers(0, 5) == 0:
                act = act[0]
            tag = dict(case=case, size=size, shift=shift, fading=fading, pad=pad, N=N, shape=act.shape)
            res = {}
            for side, fn in (('reference', ref_fn), ('oracle', oracle.activity_time_to_frequency)):
                try:
                    res[side] = np.asarray(fn(act, size, shift, fading, stft_pad=pad))
                except Exception as e:
                    res[side] = type(e).__name__
            r, o = res['reference'], res['oracle']
            if isinstance(r, str) or isinstance(o, str):
                if not (isinstance(r, str) and isinstance(o, str)):
                    print('only one raises', r if isinstance(r, str) else 'ok', o if isinstance(o, str) else 'ok', tag)
                    bad += 1
                continue
            if r.shape != o.shape or r.dtype != o.dtype or not np.array_equal(r, o):
                print('differs', r.shape, o.shape, r.dtype, o.dtype, tag)
                bad += 1
    print('activity fuzz: seed', seed, 'cases', cases, 'failures', bad)


if __name__ == '__main__':
    main()
