#!/bin/bash
# A/B of library builds on ONE box (clocks differ between boxes by a few per cent): tools/ab_bench.sh WORKLOAD REPS LIB_A LIB_B ...
# ("" or "default" = the shipped library); prints lines-kernel ms per step of every repetition, the builds interleaved.
W=$1; R=$2; shift 2
mkdir -p gpurun_out/ab
for r in $(seq 1 $R); do
  for lib in "$@"; do
    if [ "$lib" = "default" ] || [ -z "$lib" ]; then unset MONORTM_HIP_LIB; else export MONORTM_HIP_LIB=$PWD/$lib; fi
    python bench.py --workload $W --steps 100 --no-extra --no-cpu-baseline --no-pmc --detail-file gpurun_out/ab/d.json 2>gpurun_out/ab/err.txt >/dev/null || { tail -2 gpurun_out/ab/err.txt | cut -c1-200; continue; }
    python - "$lib" <<'P'
import json,sys
j=json.load(open("gpurun_out/ab/d.json")); print(f'{sys.argv[1]:40s} step {j["ms_per_step"]:.4f} lines {j["kernel_ms_per_step"]["lines"]:.4f} finish {j["kernel_ms_per_step"]["continuum_cloud_total"]:.4f} rtm {j["kernel_ms_per_step"]["rtm"]:.4f}')
P
  done
done
