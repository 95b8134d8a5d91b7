"""profiles/traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; TCC cannot
hold both in one pass) of `python bench.py --steps 2 --warmup 1`.

usage: python tools/pmc_traffic.py <dir with pmc_FETCH_SIZE/ and pmc_WRITE_SIZE/> <out.json>

Units as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes: both counters are in
KiB-sized units of 1024 B here (rocprofv3 reports FETCH_SIZE / WRITE_SIZE in kilobytes); on
gfx950 FETCH_SIZE counts 128-B requests at 64 B, so fetch bytes are doubled.  Per launch =
total over the launches of a kernel / number of launches.  Kernel names are mapped to the labels
bench.py uses.
"""
import collections, csv, glob, json, re, sys

LABELS = [
    ('wpe_corr_persist_kernel', 'wpe_corr'), ('wpe_corr_dma_kernel', 'wpe_corr'), ('wpe_corr_kernel', 'wpe_corr'), ('wpe_apply_kernel', 'wpe_apply'), ('wpe_power_kernel', 'wpe_power'),
    ('chol_update_kernel', 'wpe_chol_update'), ('chol_diag_kernel', 'wpe_chol_diag'),
    ('chol_trsm_kernel', 'wpe_chol_trsm'), ('chol_backsolve_kernel', 'wpe_backsolve'),
    ('em_estep_reg_kernel', 'em_estep'), ('em_estep_kernel', 'em_estep_first'),
    ('em_prepare_kernel', 'em_prepare'), ('em_chol_kernel', 'em_chol'), ('em_eigh_kernel', 'em_eigh'),
    ('em_onchip4_kernel', 'em_onchip'),
    ('mstep_reg_kernel', 'em_mstep'), ('stft_kernel', 'stft'), ('istft_frames_kernel', 'istft_frames'),
    ('istft_ola_kernel', 'istft_ola'), ('mvdr_solve_kernel', 'mvdr_solve'), ('mvdr_apply_kernel', 'mvdr_apply'),
    ('mvdr_ref_kernel', 'mvdr_ref'),
]


def label(kernel_name):
    if 'wcov_kernel' in kernel_name:
        # wcov_kernel<K, NORMALISE, SRC_FDT, ...>: the PSD matrices run on raw observations
        m = re.search(r'wcov_kernel<(\d+), *(\w+), *(\w+)', kernel_name)
        return 'em_mstep' if m and m.group(3) in ('true', '1') else 'psd'
    if 'em_estep_reg_kernel' in kernel_name:
        m = re.search(r'em_estep_reg_kernel<(\d+), *(\d+), *(\d+)', kernel_name)
        return 'em_predict' if m and m.group(3) == '2' else 'em_estep'
    for key, lab in LABELS:
        if key in kernel_name:
            return lab
    m = re.search(r'(\w+)_kernel', kernel_name)
    return m.group(1) if m else kernel_name[:24]


def totals(directory, counter):
    acc, n = collections.defaultdict(float), collections.defaultdict(set)
    for path in glob.glob(f'{directory}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(path)):
            if r['Counter_Name'] != counter:
                continue
            lab = label(r['Kernel_Name'])
            acc[lab] += float(r['Counter_Value'])
            n[lab].add((path, r['Dispatch_Id']))
    return {k: acc[k] / len(n[k]) for k in acc}


def main():
    base, out = sys.argv[1], sys.argv[2]
    fetch = totals(f'{base}/pmc_FETCH_SIZE', 'FETCH_SIZE')
    write = totals(f'{base}/pmc_WRITE_SIZE', 'WRITE_SIZE')
    res = {}
    for k in sorted(set(fetch) | set(write)):
        f = 2.0 * 1024.0 * fetch.get(k, 0.0)
        w = 1024.0 * write.get(k, 0.0)
        res[k] = {'fetch_bytes': f, 'write_bytes': w, 'bytes': f + w}
    # what the counters were taken on: bench.py drops `roofline.traffic` (marks it stale) when
    # the sources of the dominant kernel's translation unit hash differently
    import os
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
    from pb_chime5_amd import roofline
    res['__meta__'] = {'source_hashes': roofline.source_hashes(),
                       'gss_variant': os.environ.get('GSS_VARIANT', '')}
    json.dump(res, open(out, 'w'), indent=1)
    for k, v in sorted(((k, v) for k, v in res.items() if k != '__meta__'),
                       key=lambda kv: -kv[1]['bytes'])[:12]:
        print(f"{k:18s} fetch {v['fetch_bytes']/1e6:9.1f} MB  write {v['write_bytes']/1e6:9.1f} MB per launch")


if __name__ == '__main__':
    main()
