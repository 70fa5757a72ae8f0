#!/bin/bash
# build_dbg/libmonortm_hip_<tag>.so with extra -D flags:  tools/build_variant.sh TAG -DFOO -DBAR ...
set -e
TAG=$1; shift
cd "$(dirname "$0")/../monortm_amd/csrc"
mkdir -p ../../build_dbg
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -Wno-pass-failed -Wno-unused-const-variable -Wno-dangling-else "$@" \
  -o ../../build_dbg/libmonortm_hip_$TAG.so api.hip lines_kernel.hip continuum_kernel.hip xsec_kernel.hip rtm_kernel.hip line_table.cpp 2>&1 | grep -v "warning\|^ *[0-9]* |\|\^\|generated" || true
ls -la ../../build_dbg/libmonortm_hip_$TAG.so
