# A/B of the correlation kernel's queue order: time per launch and L2 -> fabric fetch
# (round 5: the GSS_CORR_QBLOCK / GSS_CORR_FMAJOR switches this script drove were removed from the
# library after their A/B runs -- profiles/r04*; the script documents how they were measured)
# (rocprofv3 --pmc FETCH_SIZE, raw units = 64 B x 1/2) per launch.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in "GSS_CORR_QBLOCK=0" "GSS_CORR_QBLOCK=4" "GSS_CORR_QBLOCK=8" "GSS_CORR_QBLOCK=16" "GSS_CORR_FMAJOR=1"; do
  echo "== $v"; env $v python $R/tools/wpe_kprof.py 24 941 | tr " " "\n" | paste - - - | grep corr
  env $v python $R/tools/wpe_kprof.py 24 2169 | tr " " "\n" | paste - - - | grep corr
  rm -rf /tmp/pf; env $v rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf -o p -- python $R/tools/wpe_kprof.py 24 941 513 1 > /tmp/pf.log 2>&1
  python - <<P
import csv,glob
f=glob.glob('/tmp/pf/**/*counter_collection.csv',recursive=True)[0]
tot=0;cnt=0
for r in csv.DictReader(open(f)):
    if 'corr' in r['Kernel_Name']:
        tot+=float(r['Counter_Value']); cnt+=1
print('  FETCH_SIZE per launch: %.0f raw = %.0f MB (x 2 x 1024 B, tools/pmc_traffic.py)' % (tot/cnt, tot/cnt*2048/1e6))
P
done
