"""Per-launch time (us) of the WPE kernels for F frequencies (24 channels, 10 taps, T = 941):
does the 513th frequency -- the third workgroup of one CU -- set the kernel's duration?
    python tools/solve_by_f.py [F ...]"""
import sys
import numpy as np
sys.path.insert(0, '.')
from pb_chime5_amd import ops

ctx = ops.default_context()
for F in [int(a) for a in sys.argv[1:]] or [256, 512, 513, 768, 769]:
    rng = np.random.default_rng(F)
    Y = (rng.standard_normal((24, 941, F)) + 1j * rng.standard_normal((24, 941, F)))
    ops.wpe_dtf(Y, 10, 2, 1, ctx=ctx)
    ctx.profile_enable(True)
    ctx.profile_reset()
    for _ in range(3):
        ops.wpe_dtf(Y, 10, 2, 3, ctx=ctx)
    rep = ctx.profile_report()
    ctx.profile_enable(False)
    print(f'F = {F}: ' + ', '.join('%s %.1f' % (k[4:], v['ms'] / v['calls'] * 1e3) for k, v in rep.items() if k.startswith('wpe')))
