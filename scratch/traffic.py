import csv, re, sys, collections, json
def load(path, ctr):
    acc = collections.defaultdict(float); n = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != ctr: continue
        m = re.search(r'(\w+_kernel)(<[^>]*>)?', r['Kernel_Name'])
        name = (m.group(1) + (m.group(2) or '')) if m else r['Kernel_Name'][:30]
        acc[name] += float(r['Counter_Value']); n[name].add(r['Dispatch_Id'])
    return {k: acc[k] / len(n[k]) for k in acc}
f = load('gpurun_out/pmc_FETCH_SIZE/p_counter_collection.csv', 'FETCH_SIZE')
w = load('gpurun_out/pmc_WRITE_SIZE/p_counter_collection.csv', 'WRITE_SIZE')
for k in sorted(f, key=lambda k: -f[k])[:16]:
    print(f'{k:34s} FETCH_SIZE={f[k]:12.0f} KB  WRITE_SIZE={w.get(k,0):12.0f} KB')
