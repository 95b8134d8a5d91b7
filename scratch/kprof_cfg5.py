import sys
sys.path.insert(0, '.')
from pb_chime5_amd import ops, synthetic
from pb_chime5_amd._capi import Context
ctx = Context(0); ops._prepare_windows(ctx, 1024, 256)
u = synthetic.config5(); params = ops.make_params(bss_iterations=40, bf='gev_ban')
cs = u.ex['start_orig']['original']; ce = u.ex['end']['original'] - u.ex['end_orig']['original']
r = ops.ResidentUtterance(ctx, u.obs, u.activity_array, params)
for _ in range(2): r.enqueue(u.target_index, cs, ce)
ctx.profile_enable(True); ctx.profile_reset()
for _ in range(3): r.enqueue(u.target_index, cs, ce)
prof = ctx.profile_report()
tot = sum(v['ms'] for v in prof.values())
for k, v in sorted(prof.items(), key=lambda kv: -kv[1]['ms'])[:14]:
    print(f"{k:16s} calls {v['calls']/3:5.1f} avg_ms {v['ms']/v['calls']:.4f} share {v['ms']/tot:.3f}")
print('total ms/utt', tot / 3)
