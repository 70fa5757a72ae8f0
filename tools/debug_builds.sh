#!/bin/bash
# Builds the instrumented / ablated variants of the HIP library into build_dbg/ (git-ignored, travels with gpurun):
#   libmonortm_hip_ltiming.so   -DLINES_TIMING            s_memtime stamps per stage of lines_kernel
#   libmonortm_hip_classes.so   -DLINES_CLASS_STATS       per-class statistics of the evaluate stage (global atomics: slow)
#   libmonortm_hip_timing.so    -DMW_TIMING               s_memtime stamps per stage of finish_mw_kernel
#   libmonortm_hip_abl_LOOP.so  -DMONORTM_ABLATE_LOOP     lines_kernel stops after its prologue
#   libmonortm_hip_abl_EVAL.so  -DMONORTM_ABLATE_EVAL     lines_kernel runs prologue + prepare stages only
# Select one with MONORTM_HIP_LIB=$PWD/build_dbg/<lib> (monortm_amd/api.py).  The numbers of DESIGN.md section 3 / 5 that
# are not in bench.py's JSON line come from tools/stage_timing.py and tools/ablation_pmc.sh on these builds.
set -e
cd "$(dirname "$0")/../monortm_amd/csrc"
OUT=../../build_dbg
mkdir -p $OUT
SRC="api.hip lines_kernel.hip lines_ms_kernel.hip far_kernel.hip continuum_kernel.hip xsec_kernel.hip rtm_kernel.hip line_table.cpp"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -Wno-pass-failed -Wno-unused-const-variable -DMONORTM_EXPERIMENT=1"
/opt/rocm/bin/hipcc $FLAGS -DLINES_TIMING -o $OUT/libmonortm_hip_ltiming.so $SRC
/opt/rocm/bin/hipcc $FLAGS -DLINES_CLASS_STATS -o $OUT/libmonortm_hip_classes.so $SRC
/opt/rocm/bin/hipcc $FLAGS -DMW_TIMING -o $OUT/libmonortm_hip_timing.so $SRC
/opt/rocm/bin/hipcc $FLAGS -DMONORTM_ABLATE_LOOP -o $OUT/libmonortm_hip_abl_LOOP.so $SRC
/opt/rocm/bin/hipcc $FLAGS -DMONORTM_ABLATE_EVAL -o $OUT/libmonortm_hip_abl_EVAL.so $SRC
ls -la $OUT/*.so
