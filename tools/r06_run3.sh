cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06c
AB_ARGS="--workload 1a" AB_STEPS=20 timeout 900 bash tools/ab_env.sh "GSS_VARIANT=corr_ksplit=2" "" "GSS_VARIANT=corr_ksplit=8" "GSS_VARIANT=corr_ksplit=8,corr_stg8" > gpurun_out/r06c/ab_1a_ksplit8.txt 2>&1
cat gpurun_out/r06c/ab_1a_ksplit8.txt | head -8
