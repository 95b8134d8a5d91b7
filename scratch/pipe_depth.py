"""Session-mode throughput (UtterancePipeline, PCM16 upload + result download) against the number
of utterances in flight, config 2."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from pb_chime5_amd import ops, synthetic
u = synthetic.config2()
params = ops.make_params()
cs = u.ex['start_orig']['original']; ce = u.ex['end']['original'] - u.ex['end_orig']['original']
pcm = np.clip(np.rint(u.obs / np.abs(u.obs).max() * 30000), -32768, 32767).astype(np.int16)
for depth in (1, 2, 3, 4):
    pipe = ops.UtterancePipeline(params, depth=depth)
    def run(n):
        for i in range(n):
            if pipe.full(): pipe.pop()
            pipe.enqueue(i, pcm, u.activity_array, u.target_index, cs, ce)
        while len(pipe): pipe.pop()
    run(depth + 1)
    t = time.perf_counter(); n = 24; run(n); dt = time.perf_counter() - t
    print(f'depth {depth}: {1e3 * dt / n:.2f} ms per utterance, {n * u.seconds / dt:.0f} utterance-s/s')
    pipe.close()
