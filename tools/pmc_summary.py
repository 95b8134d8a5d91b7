import csv, re, sys, collections
path = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
meta = {}
for r in csv.DictReader(open(path)):
    m = re.search(r'(\w+_kernel)(<[^>]*>)?', r['Kernel_Name'])
    name = (m.group(1) + (m.group(2) or '')) if m else r['Kernel_Name'][:30]
    acc[name][r['Counter_Name']] += float(r['Counter_Value'])
    cnt[name].add(r['Dispatch_Id'])
    meta[name] = (r['VGPR_Count'], r['SGPR_Count'], r['LDS_Block_Size'], r['Workgroup_Size'])
names = sorted(acc, key=lambda n: -acc[n].get('SQ_WAVE_CYCLES', 0))
ctrs = sorted({c for n in acc for c in acc[n]})
print('kernel'.ljust(34), 'n', 'vgpr sgpr lds wg |', ' '.join(c.replace('SQ_', '')[:14].rjust(14) for c in ctrs))
for n in names[:14]:
    wc = acc[n].get('SQ_WAVE_CYCLES', 1) or 1
    row = []
    for c in ctrs:
        v = acc[n][c]
        row.append((f'{v/wc:.3f}' if c != 'SQ_WAVE_CYCLES' else f'{v/len(cnt[n]):.3g}').rjust(14))
    print(n[:34].ljust(34), len(cnt[n]), *meta[n], '|', ' '.join(row))
