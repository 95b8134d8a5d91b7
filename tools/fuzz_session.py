"""A session of utterances whose shapes change from one to the next (2 - 28 channels, 2 - 10
classes, 0.4 - 8 s) through ops.UtterancePipeline with 1 - 3 utterances in flight: every output
must equal, bit for bit, the one the same utterance gives on a fresh context of its own.
    python tools/fuzz_session.py [SEED] [ITEMS] [BEAMFORMER]"""
import sys
from pathlib import Path

import numpy as np

R = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(R))


def main():
    from pb_chime5_amd import _capi, ops, synthetic
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    items = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    rng = np.random.default_rng(seed)
    bf = ['mvdrSouden_ban', 'gev_ban', 'sum'][int(rng.integers(0, 3))]
    if len(sys.argv) > 3:
        bf = sys.argv[3]
    params = ops.make_params(wpe=True, wpe_taps=int(rng.integers(1, 5)), wpe_iterations=2,
                             bss_iterations=int(rng.integers(2, 6)), bss_iterations_post=int(rng.integers(0, 3)),
                             bf=bf, postfilter=[None, 'mask_mul'][int(rng.integers(0, 2))])
    utts = []
    for i in range(items):
        D = int(rng.integers(2, 29)); K = int(rng.integers(2, 11))
        N = int(rng.integers(6000, 130000)); c = int(rng.integers(0, 3000))
        utts.append((synthetic.tiny(seed=seed * 1000 + i, num_channels=D, num_samples=N,
                                    num_speakers=K - 1, context=c, noise=5e-2), c))
    want = []
    for u, c in utts:
        ctx = _capi.Context()
        try:
            want.append(ops.enhance_observation(u.obs, u.activity_array, u.target_index, c, c,
                                                params=params, ctx=ctx))
        except Exception as e:
            want.append(type(e))
        ctx.close()
    bad = 0
    for depth in (1, 2, 3):
        pipe = ops.UtterancePipeline(params, depth=depth)
        got = {}

        def pop():
            try:
                tag, x = pipe.pop()
                got[tag] = x
            except Exception as e:                       # the reference aborts this utterance
                got[len(got)] = type(e)

        order = []
        for i, (u, c) in enumerate(utts):
            if pipe.full():
                pop()
            pipe.enqueue(i, u.obs, u.activity_array, u.target_index, c, c)
        while len(pipe):
            pop()
        pipe.close()
        for i, w in enumerate(want):
            g = got.get(i)
            same = (g is w) if isinstance(w, type) else (isinstance(g, np.ndarray) and np.array_equal(g, w, equal_nan=True))
            if not same:
                print('depth', depth, 'item', i, 'differs', utts[i][0].obs.shape, type(g), type(w))
                bad += 1
    print('session fuzz: seed', seed, items, 'items, bf', bf, 'failures', bad,
          'aborted utterances', sum(isinstance(w, type) for w in want))


if __name__ == '__main__':
    main()
