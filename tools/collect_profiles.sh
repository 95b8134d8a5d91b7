#!/bin/bash
# Runs on the GPU box (gpurun): the bench line, the rocprofv3 kernel statistics of the same
# command and the --pmc passes behind profiles/ (each counter group in its own run, with
# --kernel-trace only).  Output under gpurun_out/; summaries: tools/pmc_traffic.py,
# tools/pmc_mfma.py, tools/pmc_summary.py.
#   tools/collect_profiles.sh TAG                      the headline (config 2) + the full default line
#   tools/collect_profiles.sh TAG --workload 5         another BASELINE shape (5, 3i, 1a): no full line
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r03c}
shift
W="$*"
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
bench() { python $R/bench.py --only-headline $W "$@"; }
: > $O/bench_$TAG.err
rm -rf $O/prof_stats_$TAG $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmcM $O/pmcA $O/pmcB
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats_$TAG -o $TAG -- python $R/bench.py --only-headline $W > $O/prof_stats_$TAG.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o p -- python $R/bench.py --only-headline $W --steps 2 --warmup 1 > $O/pmc_$c.log 2>&1
done
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 --kernel-trace --output-format csv \
  -d $O/pmcM -o p -- python $R/bench.py --only-headline $W --steps 2 --warmup 1 > $O/pmcM.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES \
  --kernel-trace --output-format csv -d $O/pmcA -o p -- python $R/bench.py --only-headline $W --steps 1 --warmup 1 > $O/pmcA.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VMEM SQ_WAVE_CYCLES \
  --kernel-trace --output-format csv -d $O/pmcB -o p -- python $R/bench.py --only-headline $W --steps 1 --warmup 1 > $O/pmcB.log 2>&1
cd $R
find $O/prof_stats_$TAG -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_$TAG.csv \;
python tools/pmc_traffic.py $O $O/traffic_$TAG.json
# The bench lines come LAST: `roofline.traffic` is read from profiles/traffic*.json and only
# accepted when that file was measured on the sources of the loaded library -- so the figure of
# THIS run is put where bench.py looks (on the box's copy of the tree; copy it back by hand).
case "$W" in
  "") TF=traffic.json ;;
  *) TF=traffic_$(echo $W | sed 's/.*--workload \([^ ]*\).*/\1/').json ;;
esac
cp $O/traffic_$TAG.json $R/profiles/$TF
# the headline alone: the command every profile above was taken with (back-to-back launches of
# ONE stream and ONE shape, so that per-kernel averages mean something) ...
(cd /tmp && bench) > $O/bench_${TAG}_headline.json 2>> $O/bench_$TAG.err
if [ -z "$W" ]; then
  # ... and the full default line (session mode, configs table, sharded config 3, CPU baseline)
  (cd /tmp && python $R/bench.py) > $O/bench_$TAG.json 2>> $O/bench_$TAG.err
fi
tail -n 1 $O/bench_${TAG}_headline.json | cut -c1-300
cc() { find $O/$1 -name "*counter_collection.csv" | head -1; }
kt() { find $O/$1 -name "*kernel_trace.csv" | head -1; }
python tools/pmc_mfma.py $(cc pmcM) $(kt pmcM) > $O/mfma_util_$TAG.txt; cat $O/mfma_util_$TAG.txt
(python tools/pmc_summary.py $(cc pmcA); echo; python tools/pmc_summary.py $(cc pmcB)) > $O/sq_counters_$TAG.txt
# the raw traces are large (gpurun copies back at most 64 MiB): keep the summaries only
rm -rf $O/prof_stats_$TAG $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmcM $O/pmcA $O/pmcB
du -sh $O
