#!/bin/bash
# Round-6 evidence at the final kernels, one GPU box (the commands behind profiles/r06*).
#   tools/r06_evidence.sh            everything
#   tools/r06_evidence.sh profiles   the four profiled shapes + the session / multi-rank lines
#   tools/r06_evidence.sh fuzz       the GPU test suite + the fuzz campaign
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out
mkdir -p $O
WHAT=${1:-all}
if [ $WHAT = all ] || [ $WHAT = profiles ]; then
bash tools/collect_profiles.sh r06a > $O/collect_r06a.log 2>&1
bash tools/collect_profiles.sh r06b_cfg5 --workload 5 > $O/collect_r06b_cfg5.log 2>&1
bash tools/collect_profiles.sh r06b_1a --workload 1a > $O/collect_r06b_1a.log 2>&1
bash tools/collect_profiles.sh r06b_3i --workload 3i > $O/collect_r06b_3i.log 2>&1
python bench.py --gpus 8 --n1-value 1085 > $O/r06_headline_g8_shared.json 2> $O/r06_headline_g8_shared.err
python bench.py --gpus 8 --config 3 > $O/r06_config3_g8_shared.json 2> $O/r06_config3_g8_shared.err
python bench.py --gpus 8 --config 4s > $O/r06_config4s_g8_shared.json 2> $O/r06_config4s_g8_shared.err
python bench.py --config 4s > $O/r06_config4s_g1.json 2> $O/r06_config4s_g1.err
python bench.py --config 4s --multiarray False > $O/r06_config4s_g1_one_array.json 2> $O/r06_config4s_g1_one_array.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 10 --only-headline > $O/r06_torchrun_g2_shared.json 2> $O/r06_torchrun_g2_shared.err
tail -c 300 $O/bench_r06a_headline.json
fi
if [ $WHAT = all ] || [ $WHAT = fuzz ]; then
python -m pytest tests -m gpu -x -q 2>&1 | tail -n 3 > $O/r06_gputest.txt
cat $O/r06_gputest.txt
bash tools/fuzz_all.sh r06 > $O/fuzz_r06.log 2>&1
tail -n 2 $O/fuzz_r06/summary.txt
fi
