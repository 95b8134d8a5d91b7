import sys
sys.path.insert(0, '.')
from pb_chime5_amd import ops, synthetic
from pb_chime5_amd._capi import Context
ctx = Context(0); ops._prepare_windows(ctx, 1024, 256)
D = int(sys.argv[1]) if len(sys.argv) > 1 else 4
u = synthetic.config3_item(0, num_channels=D); params = ops.make_params()
cs = u.ex['start_orig']['original']; ce = u.ex['end']['original'] - u.ex['end_orig']['original']
r = ops.ResidentUtterance(ctx, u.obs, u.activity_array, params)
for _ in range(2): r.enqueue(u.target_index, cs, ce)
ctx.profile_enable(True); ctx.profile_reset()
for _ in range(3): r.enqueue(u.target_index, cs, ce)
prof = ctx.profile_report()
tot = sum(v['ms'] for v in prof.values())
for k, v in sorted(prof.items(), key=lambda kv: -kv[1]['ms'])[:12]:
    print(f"{k:16s} calls {v['calls']/3:5.1f} avg_ms {v['ms']/v['calls']:.4f} share {v['ms']/tot:.3f}")
print(f'D={D} T={r.T} total ms/utt', round(tot / 3, 2), 'x real time', round(u.seconds / (tot / 3e3)))
