#!/bin/bash
# Counter passes for the EM kernels (each group in its own run, --kernel-trace only).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd /tmp && export TMPDIR=/tmp
rm -rf $O/pmcE1 $O/pmcE2 $O/pmcE3
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SMEM SQ_IFETCH \
  --kernel-trace --output-format csv -d $O/pmcE1 -o p -- python $R/bench.py --only-headline --steps 1 --warmup 1 > $O/pmcE1.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_IFETCH_LEVEL SQ_BUSY_CYCLES SQ_WAVES \
  --kernel-trace --output-format csv -d $O/pmcE2 -o p -- python $R/bench.py --only-headline --steps 1 --warmup 1 > $O/pmcE2.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LEVEL_WAVES SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES \
  --kernel-trace --output-format csv -d $O/pmcE3 -o p -- python $R/bench.py --only-headline --steps 1 --warmup 1 > $O/pmcE3.log 2>&1
cd $R
cc() { find $O/$1 -name "*counter_collection.csv" | head -1; }
(for p in pmcE1 pmcE2 pmcE3; do python tools/pmc_summary.py $(cc $p); echo; done) > $O/em_counters.txt
rm -rf $O/pmcE1 $O/pmcE2 $O/pmcE3
cat $O/em_counters.txt
