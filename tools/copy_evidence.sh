#!/bin/bash
# Copies what tools/r06_evidence.sh left under gpurun_out/ (scratch) to profiles/ (tracked).
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out; P=$R/profiles
for t in r06a r06b_cfg5 r06b_1a r06b_3i; do
  [ -f $O/bench_$t.json ] && cp $O/bench_$t.json $P/${t}_bench.json
  cp $O/bench_${t}_headline.json $P/${t}_bench_headline.json
  for k in kernel_stats:csv mfma_util:txt sq_counters:txt traffic:json; do
    cp $O/${k%%:*}_$t.${k##*:} $P/${t}_${k%%:*}.${k##*:}
  done
done
cp $O/traffic_r06a.json $P/traffic.json
cp $O/traffic_r06b_cfg5.json $P/traffic_5.json
cp $O/traffic_r06b_1a.json $P/traffic_1a.json
cp $O/traffic_r06b_3i.json $P/traffic_3i.json
for f in r06_headline_g8_shared r06_config3_g8_shared r06_config4s_g8_shared r06_config4s_g1 r06_config4s_g1_one_array r06_torchrun_g2_shared; do
  cp $O/$f.json $P/$f.json
done
cp $O/fuzz_r06/summary.txt $P/r06_fuzz_summary.txt
[ -f $O/r06_gputest.txt ] && cp $O/r06_gputest.txt $P/r06_gputest.txt
git -C $R rev-parse HEAD > $P/r06_evidence_head.txt
