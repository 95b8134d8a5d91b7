#!/bin/bash
# Host-side fuzzers against the REFERENCE'S OWN code (needs /root/reference: build container
# only); tallies into profiles/TAG_fuzz_reference_summary.txt.
R=$(cd "$(dirname "$0")/.." && pwd)
TAG=${1:-r04}
S=$R/profiles/${TAG}_fuzz_reference_summary.txt
echo "fuzzers against the reference's own functions and front doors, $TAG: $(git -C $R rev-parse --short HEAD)" > $S
for spec in "fuzz_solve_vs_reference.py 5 1500" "fuzz_activity_vs_reference.py 6 2000" "fuzz_rttm_vs_reference.py 7 320" "fuzz_chime5_vs_reference.py 8 90" "fuzz_chime5_vs_reference.py 9 60"; do
  set -- $spec
  echo "== tests/golden/$1 seed $2 cases $3" >> $S
  python $R/tests/golden/$1 $2 $3 2>&1 | tail -1 >> $S
done
cat $S
