#!/bin/bash
# The fuzz campaign behind DESIGN.md section 4, reproducible: every sweep with fixed seeds and
# case counts, the one-line tallies collected into ONE tracked summary
# (gpurun_out/fuzz_TAG/summary.txt -> copy to profiles/TAG_fuzz_summary.txt).  Runs on a GPU box:
#     gpurun --timeout 2400 -- 'bash tools/fuzz_all.sh r04'
# The host-side fuzzers against the REFERENCE'S OWN code (tests/golden/fuzz_*_vs_reference.py)
# need /root/reference and run in the build container: tools/fuzz_reference.sh.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r04}
O=$R/gpurun_out/fuzz_$TAG
mkdir -p $O
cd $R
S=$O/summary.txt
echo "fuzz campaign $TAG: $(git -C $R rev-parse --short HEAD 2>/dev/null || echo 'snapshot') on $(date -u +%Y-%m-%dT%H:%MZ)" > $S
run() { # name, command...
  local name=$1; shift
  echo "== $name: $*" >> $S
  ( "$@" > $O/$name.log 2>&1; echo "exit status $?" >> $O/$name.log )
  grep -E "fuzz:|fuzz: seed|failures|passed|failed|error|exit status|reference-channel mismatches" $O/$name.log | tail -4 >> $S
}
# whole pipeline, wide parameter space (tests/fuzz_params.py), every case also through the
# block-by-block orchestration
for seed in 4101 4102 4103; do
  run pipeline_$seed env GSS_FUZZ_SEED=$seed GSS_FUZZ_CASES=250 GSS_FUZZ_WIDE=1 timeout 900 \
      python -m pytest tests/test_gpu_pipeline.py -m gpu -q -s -k test_random_shapes_against_oracle
done
run stft python tools/fuzz_stft.py 41 600
run wpe python tools/fuzz_wpe.py 42 400
run wpe_one_array env GSS_FUZZ_D=4 python tools/fuzz_wpe.py 43 200        # P folded into R's last column tile
run wpe_24 env GSS_FUZZ_D=24 python tools/fuzz_wpe.py 44 60               # persistent LDS-DMA correlation
run wpe_20 env GSS_FUZZ_D=20 python tools/fuzz_wpe.py 45 60
run wpe_12 env GSS_FUZZ_D=12 python tools/fuzz_wpe.py 53 100              # outer microphones: fine tiles, 4 waves share a window
run wpe_10 env GSS_FUZZ_D=10 python tools/fuzz_wpe.py 54 100
run wpe_one_array_single_waves env GSS_FUZZ_D=4 GSS_VARIANT=corr_ksplit=1 python tools/fuzz_wpe.py 55 100
run bf python tools/fuzz_bf.py 46 600
run em python tools/fuzz_em.py 47 700
run em_one_array env GSS_FUZZ_D=4 GSS_FUZZ_KMAX=6 GSS_FUZZ_TMAX=1200 python tools/fuzz_em.py 48 500   # em_onchip4_kernel
run em_blocks env GSS_VARIANT=em_l3_fit_mb=0,em_l3_mb=1 python tools/fuzz_em.py 56 200          # EM over blocks of frequencies, two streams
run em_zero_frames env GSS_FUZZ_ZEROS=1 python tools/fuzz_em.py 57 300        # digital silence: the eigenvalue-normalised update
run em_zero_frames_one_array env GSS_FUZZ_ZEROS=1 GSS_FUZZ_D=4 GSS_FUZZ_KMAX=6 GSS_FUZZ_TMAX=1200 python tools/fuzz_em.py 58 200
# the wide pipeline draws with a dropped, zero-filled block in every recording
run pipeline_silence_61 timeout 1500 python tools/fuzz_silence.py 61 200
run pipeline_silence_62 timeout 1500 python tools/fuzz_silence.py 62 200
for seed in 49 50 51; do run session_$seed python tools/fuzz_session.py $seed 24; done
run session_gev python tools/fuzz_session.py 52 24 gev_ban
# machine-written total: every "failures N" / "N failed" / non-zero exit status of the sweeps above
python3 - "$S" <<'PY' >> $S
import re, sys
text = open(sys.argv[1]).read()
sweeps = len(re.findall(r'^== ', text, re.M))
failures = sum(int(n) for n in re.findall(r'failures (\d+)', text))
failed_tests = sum(int(n) for n in re.findall(r'(\d+) failed', text))
passed_tests = sum(int(n) for n in re.findall(r'(\d+) passed', text))
cases = sum(int(n) for n in re.findall(r'cases (\d+)', text))
bad_exit = len([s for s in re.findall(r'exit status (\d+)', text) if s != '0'])
mism = sum(len([x for x in body.split(',') if x.strip()])
           for body in re.findall(r'reference-channel mismatches: \[(.*?)\]', text))
total = failures + failed_tests + bad_exit
print(f'TOTAL: sweeps {sweeps}, stage cases {cases}, pytest cases passed {passed_tests}, failures {failures}, '
      f'failed tests {failed_tests}, non-zero exits {bad_exit}, reference-channel mismatches {mism} '
      f'=> {"CLEAN" if total == 0 else "NOT CLEAN: " + str(total)}')
PY
cat $S
