"""A/B timing of library variants: runs `bench.py --only-headline` once per library given on
the command line (paths, or 'default') and prints value + the per-kernel averages side by side.

    python tools/ab_kernels.py default pb_chime5_amd/lib/variants/libgss_kt128.so ...
"""
import json
import os
import subprocess
import sys
from pathlib import Path

R = Path(__file__).resolve().parents[1]
rows = {}
names = []
for lib in sys.argv[1:]:
    env = dict(os.environ)
    tag = 'default'
    if lib != 'default':
        spec, *envs = lib.split(',')
        tag = Path(spec).stem.replace('libgss_', '') if spec != 'default' else 'default'
        if spec != 'default':
            env['GSS_HIP_LIBRARY'] = str((R / spec).resolve())
        for e in envs:
            k, v = e.split('=')
            env[k] = v
            tag += ',' + e
    out = subprocess.run([sys.executable, str(R / 'bench.py'), '--only-headline', '--steps', '10'],
                         capture_output=True, text=True, env=env)
    try:
        d = json.loads(out.stdout.strip().splitlines()[-1])
    except Exception:
        print(tag, 'FAILED', out.stderr[-800:])
        continue
    names.append(tag)
    rows.setdefault('VALUE', {})[tag] = d['value']
    for k, v in d['kernels'].items():
        rows.setdefault(k, {})[tag] = v['avg_ms'] * v['calls_per_step']
print('%-18s' % 'ms/utt' + ''.join('%22s' % n[-21:] for n in names))
for k, r in rows.items():
    print('%-18s' % k + ''.join('%22.4f' % r.get(n, float('nan')) for n in names))
