"""What a block of digital silence costs: the config-2 utterance (24 channels, 15 s) and a
dev-shaped one-array item through the fused pipeline, as they are and with one second of every
channel zeroed -- frequencies that hold all-zero frames take the eigendecomposition for every
class in every EM iteration (EXPERIMENTS round 6, item 12).
    python tools/silence_cost.py"""
import sys
import time
from pathlib import Path

import numpy as np

R = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(R))


def main():
    from pb_chime5_amd import ops, synthetic
    u2 = synthetic.config2()
    n3 = 554490
    u1 = synthetic.make_utterance(1001, 4, n3, [(240000, n3 - 240000), (250000, 300000), (100000, 500000)],
                                  start_context=240000, end_context=240000)
    for name, u in (('config 2 (24 channels, 15 s)', u2), ('one array (4 channels, 34.7 s)', u1)):
        cs = u.ex['start_orig']['original']
        ce = u.ex['end']['original'] - u.ex['end_orig']['original']
        for label, a, b in (('as it is', 0, 0), ('1 s of zeros in every channel', 5, 6)):
            obs = np.array(u.obs)
            obs[:, a * 16000:b * 16000] = 0.0
            ms = []
            for _ in range(4):
                t0 = time.perf_counter()
                ops.enhance_observation(obs, u.activity_array, u.target_index, cs, ce)
                ms.append(1e3 * (time.perf_counter() - t0))
            print(f'{name}: {label}: {min(ms[1:]):.1f} ms per utterance (host buffers in and out)', flush=True)


if __name__ == '__main__':
    main()
