"""How good is the bin-sampled CPU baseline of bench.py?  One config-2 utterance (BASELINE.json
configs[1]) through the oracle on ONE core with ALL 513 frequency bins, next to the figure
bench.py extrapolates from 24 sampled bins on the same machine.

    python tools/cpu_full_utterance.py profiles/r03_cpu_full_utterance.json

Offline (minutes of CPU); the result is committed under profiles/."""
import json
import os
import sys
import time
from pathlib import Path

for var in ('OMP_NUM_THREADS', 'MKL_NUM_THREADS', 'OPENBLAS_NUM_THREADS'):
    os.environ[var] = '1'          # like /root/reference/pb_chime5/__init__.py:3-14

import numpy as np  # noqa: E402

R = Path(__file__).resolve().parents[1]
for p in (str(R), str(R / 'oracle')):
    if p not in sys.path:
        sys.path.insert(0, p)

import bench  # noqa: E402
import gss_oracle as oracle  # noqa: E402
from pb_chime5_amd import synthetic  # noqa: E402


def main():
    u = synthetic.config2()
    t0 = time.perf_counter()
    oracle.enhance_observation(u.obs, u.activity_array, u.target_index, u.ex)
    full = time.perf_counter() - t0
    # the sampled estimate, computed exactly like a bench.py worker does
    import tempfile
    utt1 = synthetic.config1()
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, 'w.npz')
        np.savez(path, obs1=utt1.obs, act1=utt1.activity_array.astype(np.uint8), obs2=u.obs,
                 act2=u.activity_array.astype(np.uint8), ctx2=u.ex['start_orig']['original'])
        est = bench._cpu_worker((path, 24, 1))
    model, physical, logical, mem_gb, quota = bench._host_cpu()
    out = {
        'what': 'one config-2 utterance (24 ch, 15 s, 10 taps, 3 WPE + 20 EM iterations, MVDR+BAN) '
                'through oracle.enhance_observation on one core, all 513 bins, against the '
                'bench.py estimate from 24 sampled bins on the same core',
        'cpu_model': model, 'full_513_bins_s': full, 'sampled_estimate_s': est['cfg2_s'],
        'sampled_parts_s': {'full_part': est['cfg2_full_part_s'], 'bins_part_24': est['cfg2_bins_part_s']},
        'estimate_over_full': est['cfg2_s'] / full,
        'per_core_utterance_seconds_per_s': {'full': u.seconds / full, 'estimate': u.seconds / est['cfg2_s']},
    }
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 1:
        Path(sys.argv[1]).write_text(json.dumps(out, indent=1) + '\n')


if __name__ == '__main__':
    main()
