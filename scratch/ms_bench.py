import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from pb_chime5_amd import ops, synthetic
from pb_chime5_amd._capi import Context
S = int(sys.argv[1]); steps = int(sys.argv[2])
utt = synthetic.config2()
ctxs = [Context(0) for _ in range(S)]
params = ops.make_params()
res = []
for c in ctxs:
    ops._prepare_windows(c, 1024, 256)
    res.append(ops.ResidentUtterance(c, utt.obs, utt.activity_array, params))
cs = utt.ex['start_orig']['original']
for r in res:
    r.enqueue(0, cs, cs)
for c in ctxs: c.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    for r in res:
        r.enqueue(0, cs, cs)
for c in ctxs: c.synchronize()
el = time.perf_counter() - t0
print(f'streams={S} utt/s={S*steps/el:.2f} rtf={S*steps*15/el:.1f} ms/utt={1e3*el/(S*steps):.2f}')
