#!/bin/bash
# tools/build_variant.sh NAME [-DFOO=1 ...]: an experimental build of libgss_hip.so with extra
# compiler flags, written to pb_chime5_amd/lib/variants/libgss_NAME.so (select it with
# GSS_HIP_LIBRARY=<path>).  For A/B timing of kernel variants on the GPU box.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
OUT=$R/pb_chime5_amd/lib/variants; mkdir -p $OUT/$NAME
FLAGS="-DGSS_EXPERIMENT_BUILD=1 --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -fno-fast-math -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form=1"
for f in gss_api stft wpe cacgmm mvdr; do
  /opt/rocm/bin/hipcc $FLAGS "$@" -c $R/pb_chime5_amd/csrc/$f.hip -o $OUT/$NAME/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OUT/$NAME/*.o -o $OUT/libgss_$NAME.so
rm -rf $OUT/$NAME
echo $OUT/libgss_$NAME.so
