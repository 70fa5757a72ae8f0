#!/bin/bash
# One A/B variant of the HIP library for lines_ms_kernel: lines_ms_kernel.hip and api.hip recompiled with extra flags (experiment
# switches allowed), the other objects from the shipped build (monortm_amd/lib/obj/).
#   tools/build_variant_ms.sh NAME [-DFLAG ...]   -> build_dbg/libmonortm_hip_NAME.so   (git-ignored, travels with gpurun)
set -e
NAME=$1; shift
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
CSRC=$ROOT/monortm_amd/csrc
OBJ=$ROOT/monortm_amd/lib/obj
OUT=$ROOT/build_dbg
mkdir -p $OUT/obj_$NAME
CF="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-pass-failed -Wno-unused-const-variable"
/opt/rocm/bin/hipcc $CF -mllvm -disable-machine-licm -DMONORTM_EXPERIMENT=1 "$@" -c $CSRC/lines_ms_kernel.hip -o $OUT/obj_$NAME/lines_ms_kernel.o &
/opt/rocm/bin/hipcc $CF -DMONORTM_EXPERIMENT=1 "$@" -c $CSRC/api.hip -o $OUT/obj_$NAME/api.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libmonortm_hip_$NAME.so $OUT/obj_$NAME/lines_ms_kernel.o $OUT/obj_$NAME/api.o \
    $OBJ/lines_kernel.o $OBJ/far_kernel.o $OBJ/continuum_kernel.o $OBJ/xsec_kernel.o $OBJ/rtm_kernel.o $OBJ/line_table.o
ls -la $OUT/libmonortm_hip_$NAME.so
