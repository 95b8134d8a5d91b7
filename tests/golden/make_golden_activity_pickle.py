"""A per-session activity pickle as the REFERENCE writes it (core.py:135-139 reads
`<path>/<session>.pkl`): a dict array -> speaker -> ArrayIntervall, pickled here with the
reference's own pb_chime5/utils/intervall_array.py (its Cython helper compiled from the
reference's .pyx, build container only).  The reference's `__reduce__`
(utils/intervall_array.py:145-164) returns `self.from_str`, a staticmethod around the
MODULE-LEVEL function `ArrayIntervall_from_str` (:12,104), so the payload names
`pb_chime5.utils.intervall_array.ArrayIntervall_from_str` -- that is the name a drop-in has to
resolve (the stand-in of the earlier test named `ArrayIntervall.from_str` instead).

    python tests/golden/make_golden_activity_pickle.py

writes tests/golden/activity_reference.pkl (the bytes) and activity_reference.json (for every
track: shape, normalised intervals, and a few dense slices the reference's object returns, as [start, stop) runs)."""
import json
import pickle
import sys
import tempfile
import types
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import make_golden as mg  # noqa: E402


def _runs(dense):
    edges = np.diff(np.concatenate([[0], np.asarray(dense, dtype=np.int8), [0]]))
    return [list(map(int, r)) for r in zip(np.flatnonzero(edges > 0), np.flatnonzero(edges < 0))]


def main():
    rng = np.random.default_rng(2024)
    with tempfile.TemporaryDirectory() as tmp:
        ref = mg._prepare_reference(Path(tmp))
        sys.path.insert(0, str(ref))
        pkg = types.ModuleType('pb_chime5')
        pkg.__path__ = [str(ref / 'pb_chime5')]
        pkg.git_root = ref
        sys.modules['pb_chime5'] = pkg
        from pb_chime5.utils.intervall_array import ArrayIntervall

        n = 200000
        store, expect = {}, {}
        for array in ('U01', 'U02', 'U06'):
            store[array], expect[array] = {}, {}
            for speaker in ('P05', 'P06', 'P07', 'P08', 'Noise'):
                ai = ArrayIntervall(shape=[n])
                if speaker == 'Noise':
                    ai[0:n] = 1
                else:
                    for _ in range(int(rng.integers(1, 9))):
                        s = int(rng.integers(0, n - 10))
                        ai[s:min(n, s + int(rng.integers(1, 20000)))] = 1
                store[array][speaker] = ai
                slices = [(0, 64), (n - 64, n)] + [
                    (s, s + int(rng.integers(1, 5000)))
                    for s in map(int, rng.integers(0, n - 5000, 4))]
                expect[array][speaker] = dict(
                    shape=list(ai.shape),
                    intervals=[list(map(int, i)) for i in ai.normalized_intervals],
                    slices=[dict(start=a, stop=b, runs=_runs(ai[a:b])) for a, b in slices])
        blob = pickle.dumps(store, protocol=pickle.HIGHEST_PROTOCOL)
    assert b'ArrayIntervall_from_str' in blob and b'pb_chime5_amd' not in blob
    (HERE / 'activity_reference.pkl').write_bytes(blob)
    (HERE / 'activity_reference.json').write_text(json.dumps(expect))
    print('wrote', len(blob), 'bytes')


if __name__ == '__main__':
    main()
