import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
import numpy as np
from pb_chime5_amd import ops, synthetic
from pb_chime5_amd._capi import default_context
import gss_oracle as oracle
ctx = default_context(0)
def run(name, u, **kw):
    cs = u.ex['start_orig']['original']; ce = u.ex['end']['original'] - u.ex['end_orig']['original']
    t = time.time()
    x, det = ops.enhance_observation(u.obs, u.activity_array, u.target_index, cs, ce, debug=True, ctx=ctx, **kw)
    t1 = time.time() - t
    t = time.time(); x2 = ops.enhance_observation(u.obs, u.activity_array, u.target_index, cs, ce, ctx=ctx, **kw); t2 = time.time() - t
    print(name, u.obs.shape, 'T', det['Obs'].shape[1], 'finite', bool(np.all(np.isfinite(x))), 'ref', det['ref_channel'],
          f'{t2*1e3:.0f} ms (incl. H2D/D2H) rtf {u.seconds/t2:.0f}', 'repeat-equal', bool(np.array_equal(x, x2)), 'ws %.2f GB' % (ctx.workspace_bytes()/1e9))
    return u, x, det
u, x, det = run('config3[0]', synthetic.config3_item(0))
# spot-check EM on 3 bins against the oracle from the GPU's own dereverberated tensor
bins = [5, 200, 500]
actf = oracle.activity_time_to_frequency(u.activity_array, 1024, 256, True)
post = oracle.gss_block(det['Obs'][..., bins], actf, 20, 1)
print('  config3 EM max|dgamma| on 3 bins', np.max(np.abs(post - det['posterior'][..., bins])))
u, x, det = run('config5', synthetic.config5(), bss_iterations=40)
Y = oracle.stft(u.obs)[..., bins]
print('  config5 WPE rel err on 3 bins', np.max(np.abs(oracle.wpe_block(Y, 10, 2, 3) - det['Obs'][..., bins])) / np.max(np.abs(Y)))
u, x, det = run('config1', synthetic.config1(), wpe=False, bss_iterations=5)
