#!/bin/bash
# tools/pmc_traffic_only.sh TAG [bench args]: only the two --pmc passes (FETCH_SIZE, WRITE_SIZE)
# of tools/collect_profiles.sh and their summary -> gpurun_out/traffic_TAG.json
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o p -- python $R/bench.py --only-headline "$@" --steps 2 --warmup 1 > $O/pmc_$c.log 2>&1
done
cd $R
python tools/pmc_traffic.py $O $O/traffic_$TAG.json
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
