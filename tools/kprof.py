"""Per-kernel table (HIP events around every launch) for one synthetic shape.

    python tools/kprof.py [channels=24] [seconds=15] [em_iterations=20] [bf=mvdrSouden_ban]
"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
from pb_chime5_amd import ops, synthetic
from pb_chime5_amd._capi import default_context

D = int(sys.argv[1]) if len(sys.argv) > 1 else 24
sec = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
it = int(sys.argv[3]) if len(sys.argv) > 3 else 20
bf = sys.argv[4] if len(sys.argv) > 4 else 'mvdrSouden_ban'
n = int(sec * 16000)
ctxs = n // 3
iv = [(ctxs, n - ctxs), (n // 10, n // 2), (n // 3, n - n // 8), (n // 20, n // 4)]
u = synthetic.make_utterance(11, D, n, iv, start_context=ctxs, end_context=ctxs, rir_taps=1024,
                             noise=3e-2, fast=True)
ctx = default_context(0)
params = ops.make_params(bss_iterations=it, bf=bf)
ops._prepare_windows(ctx, 1024, 256)
res = ops.ResidentUtterance(ctx, u.obs, u.activity_array, params)
for _ in range(2):
    res.enqueue(0, ctxs, ctxs)
ctx.synchronize()
import time
t0 = time.perf_counter()
for _ in range(5):
    res.enqueue(0, ctxs, ctxs)
ctx.synchronize()
ms = (time.perf_counter() - t0) / 5 * 1e3
ctx.profile_enable(True); ctx.profile_reset()
for _ in range(3):
    res.enqueue(0, ctxs, ctxs)
prof = ctx.profile_report(); ctx.profile_enable(False)
print(f'D={D} T={res.T} K={res.K}: {ms:.3f} ms per utterance = {sec / ms * 1e3:.0f} x real time')
tot = sum(v['ms'] for v in prof.values()) / 3
for k, v in sorted(prof.items(), key=lambda kv: -kv[1]['ms']):
    print(f"  {k:18s} {v['calls'] / 3:6.1f} x {v['ms'] / v['calls']:8.4f} ms = {v['ms'] / 3:7.3f} ms  {100 * v['ms'] / 3 / tot:5.1f} %")
