import sys, time
sys.path.insert(0, '.')
import numpy as np
from pb_chime5_amd import ops, synthetic
from pb_chime5_amd._capi import Context
ctx = Context(0)
params = ops.make_params()
ops._prepare_windows(ctx, 1024, 256)
utt = synthetic.config2(seed=2)
cs = utt.ex['start_orig']['original']
res = ops.ResidentUtterance(ctx, utt.obs, utt.activity_array, params)
for prof in (False, True, False):
    ctx.profile_enable(prof)
    for _ in range(3): res.enqueue(utt.target_index, cs, cs)
    ctx.synchronize()
    t = time.perf_counter()
    for _ in range(20): res.enqueue(utt.target_index, cs, cs)
    ctx.synchronize()
    e = time.perf_counter() - t
    print('profiling', prof, 'ms/utt', round(1e3 * e / 20, 3), 'utt-s/s', round(20 * 15 / e, 1))
