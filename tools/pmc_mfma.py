"""Per-kernel MFMA utilisation and effective clock from one rocprofv3 --pmc pass
(GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64)
joined with the kernel trace of the same run (durations)."""
import csv, re, sys, collections
cc, kt = sys.argv[1], sys.argv[2]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-9
acc = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for r in csv.DictReader(open(cc)):
    m = re.search(r'(\w+_kernel)', r['Kernel_Name'])
    name = m.group(1) if m else r['Kernel_Name'][:30]
    acc[name][r['Counter_Name']] += float(r['Counter_Value'])
    disp[name].add(r['Dispatch_Id'])
print(f"{'kernel':24s} {'launches':>8s} {'avg ms':>8s} {'eff. GHz':>9s} {'MFMA busy':>10s} {'f64 MOPS/launch':>16s}")
for name in sorted(acc, key=lambda n: -sum(dur.get(d, 0) for d in disp[n]))[:8]:
    t = sum(dur.get(d, 0) for d in disp[name])
    a = acc[name]
    n = len(disp[name])
    gui = a.get('GRBM_GUI_ACTIVE', 0)
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES over the 1024
    # SIMDs, in cycles; one MFMA_MOPS_F64 = 512 flop
    gui /= 8.0
    ghz = gui / t / 1e9 if t else 0
    busy = a.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (gui * 1024) if gui else 0
    print(f"{name:24s} {n:8d} {1e3*t/n:8.4f} {ghz:9.3f} {busy:10.3f} {a.get('SQ_INSTS_VALU_MFMA_MOPS_F64', 0)/n:16.4g}")
