"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs because the
TCC block has 4 counter slots) into per-launch HBM-side traffic per kernel.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out_f -o p -- python bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out_w -o p -- python bench.py ...
    python tools/pmc_traffic.py out_f/p_counter_collection.csv out_w/p_counter_collection.csv \
        > profiles/traffic.json

Units / corrections (MI355X_MICROARCH.md, section HBM): the counters are in KiB;
on gfx950 FETCH_SIZE reports half of the bytes of wide (16 B / lane) coalesced
streaming reads -- every kernel here loads complex128 = 16 B per lane -- so the
fetch side is doubled.  Infinity-Cache hits are counted, so this is fabric
traffic, an upper bound of HBM traffic.
"""
import collections
import csv
import json
import re
import sys

BENCH_NAME = {
    'wpe_corr_kernel': 'wpe_corr', 'wpe_apply_kernel': 'wpe_apply',
    'wpe_power_kernel': 'wpe_power', 'chol_diag_kernel': 'wpe_chol_diag',
    'chol_trsm_kernel': 'wpe_chol_trsm', 'chol_update_kernel': 'wpe_chol_update',
    'chol_backsolve_kernel': 'wpe_backsolve', 'em_chol_kernel': 'em_chol',
    'em_eigh_kernel': 'em_eigh', 'stft_kernel': 'stft', 'mvdr_apply_kernel': 'mvdr_apply',
    'mvdr_solve_kernel': 'mvdr_solve', 'istft_frames_kernel': 'istft_frames',
    'istft_ola_kernel': 'istft_ola', 'masks_kernel': 'masks', 'em_prepare_kernel': 'em_prepare',
}


def bench_name(kernel):
    m = re.search(r'(\w+_kernel)(<([^>]*)>)?', kernel)
    if not m:
        return None
    base, targs = m.group(1), m.group(3)
    if base == 'em_estep_kernel':
        mode = targs.split(',')[1].strip()
        return {'0': 'em_estep_first', '1': 'em_estep', '2': 'em_predict'}[mode]
    if base == 'em_estep_reg_kernel':
        mode = targs.split(',')[2].strip()
        return {'1': 'em_estep', '2': 'em_predict'}[mode]
    if base == 'wcov_kernel':
        k = targs.split(',')[0].strip()
        return 'psd' if (k == '2' and 'false, false' in targs) else 'em_mstep'
    return BENCH_NAME.get(base)


def per_launch(path, counter):
    total = collections.defaultdict(float)
    launches = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        name = bench_name(r['Kernel_Name'])
        if name is None:
            continue
        total[name] += float(r['Counter_Value'])
        launches[name].add(r['Dispatch_Id'])
    return {k: total[k] / len(launches[k]) for k in total}


def main():
    fetch = per_launch(sys.argv[1], 'FETCH_SIZE')
    write = per_launch(sys.argv[2], 'WRITE_SIZE')
    out = {}
    for k in sorted(set(fetch) | set(write)):
        f = 2.0 * fetch.get(k, 0.0) * 1024.0       # gfx950: x2 for 16 B/lane streams
        w = write.get(k, 0.0) * 1024.0
        out[k] = {'fetch_bytes': f, 'write_bytes': w, 'bytes': f + w}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == '__main__':
    main()
