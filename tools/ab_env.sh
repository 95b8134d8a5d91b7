#!/bin/bash
# tools/ab_env.sh "ENV1=a ENV2=b" "ENV3=c" ...: bench.py --only-headline once per environment
# setting (first argument "" = default), value + per-kernel ms/utterance side by side.
# AB_ARGS="--workload 5" adds bench arguments (another workload); AB_STEPS: timed steps (10).
R=$(cd "$(dirname "$0")/.." && pwd)
i=0
for e in "$@"; do
  env $e python $R/bench.py --only-headline --steps ${AB_STEPS:-10} $AB_ARGS > /tmp/ab_$i.json 2>/tmp/ab_$i.err || tail -3 /tmp/ab_$i.err
  i=$((i+1))
done
python - "$@" <<'PY'
import json, sys
rows, names = {}, []
for i, e in enumerate(sys.argv[1:]):
    try:
        d = json.loads(open(f'/tmp/ab_{i}.json').read().strip().splitlines()[-1])
    except Exception as ex:
        print(e or 'default', 'FAILED', ex); continue
    tag = e or 'default'; names.append(tag)
    rows.setdefault('VALUE', {})[tag] = d['value']
    rows.setdefault('ms_per_step', {})[tag] = d['ms_per_step']
    for k, v in d['kernels'].items():
        rows.setdefault(k, {})[tag] = v['avg_ms'] * v['calls_per_step']
print('%-18s' % 'ms/utt' + ''.join('%26s' % n[-25:] for n in names))
for k, r in rows.items():
    print('%-18s' % k + ''.join('%26.4f' % r.get(n, float('nan')) for n in names))
PY
