cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06d
(time python tools/fuzz_em.py 47 700) > gpurun_out/r06d/fuzz_em_47.txt 2>&1
(time env GSS_FUZZ_D=4 GSS_FUZZ_KMAX=6 GSS_FUZZ_TMAX=1200 python tools/fuzz_em.py 48 500) > gpurun_out/r06d/fuzz_em_48.txt 2>&1
(time python tools/fuzz_em.py 147 600) > gpurun_out/r06d/fuzz_em_147.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_stages.py -m gpu -q -s -k "floor_decided or unknown_variant" 2>&1 | tail -6 > gpurun_out/r06d/pytest_floor.txt
for f in gpurun_out/r06d/fuzz_em_47.txt gpurun_out/r06d/fuzz_em_48.txt gpurun_out/r06d/fuzz_em_147.txt gpurun_out/r06d/pytest_floor.txt; do tail -n 6 $f; done
