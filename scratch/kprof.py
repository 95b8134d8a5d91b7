"""Per-kernel profile of one bench configuration: python scratch/kprof.py [n_top]"""
import json, subprocess, sys
out = subprocess.run([sys.executable, 'bench.py', '--steps', '6', '--warmup', '2', '--no-cpu-baseline',
                      ], capture_output=True, text=True)
line = out.stdout.strip().splitlines()[-1]
d = json.loads(line)
print('value', round(d['value'], 1), 'ms/step', round(d['ms_per_step'], 3))
top = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for k, v in list(d['kernels'].items())[:top]:
    print(f"  {k:14s} calls {v['calls_per_step']:5.1f} avg_ms {v['avg_ms']:.4f} share {v['share']:.3f}")
