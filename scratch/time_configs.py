"""Single-stream device time of BASELINE configs 2, 3 (first items) and 5, inputs resident."""
import sys, time
sys.path.insert(0, '.')
from pb_chime5_amd import ops, synthetic
from pb_chime5_amd._capi import Context
ctx = Context(0)
ops._prepare_windows(ctx, 1024, 256)
cases = [('config2', synthetic.config2(seed=2), dict()),
         ('config3[0]', synthetic.config3_item(0), dict()),
         ('config3[1]', synthetic.config3_item(1), dict()),
         ('config5 (gev_ban, 40 it)', synthetic.config5(), dict(bss_iterations=40, bf='gev_ban'))]
for name, u, kw in cases:
    params = ops.make_params(**kw)
    cs = u.ex['start_orig']['original']; ce = u.ex['end']['original'] - u.ex['end_orig']['original']
    r = ops.ResidentUtterance(ctx, u.obs, u.activity_array, params)
    for _ in range(2): r.enqueue(u.target_index, cs, ce)
    ctx.synchronize()
    n = 5
    t = time.perf_counter()
    for _ in range(n): r.enqueue(u.target_index, cs, ce)
    ctx.synchronize()
    e = (time.perf_counter() - t) / n
    print(f'{name:26s} D={r.D:2d} T={r.T:5d} {u.seconds:6.1f} s audio  {1e3*e:8.2f} ms  {u.seconds/e:8.1f} x real time  workspace {ctx.workspace_bytes()/1e9:.2f} GB')
