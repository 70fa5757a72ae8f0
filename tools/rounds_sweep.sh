#!/bin/bash
# lines_ms_kernel against lines_kernel over batch sizes (rounds of waves) on one box: tools/rounds_sweep.sh "128 192 256 320 384 512 768" [WORKLOAD = c4shard, c2lc, c4brd]
mkdir -p gpurun_out/ab
for n in $1; do for k in wn ms auto; do
  MONORTM_LINES_KERNEL=$k python bench.py --workload ${2:-c4shard} --profiles-per-gpu $n --steps 100 --no-extra --no-cpu-baseline --no-pmc --detail-file gpurun_out/ab/d.json 2>gpurun_out/ab/err.txt >/dev/null || { tail -2 gpurun_out/ab/err.txt | cut -c1-200; continue; }
  python - $n $k <<'P'
import json,sys
j=json.load(open("gpurun_out/ab/d.json")); print(f'{sys.argv[1]:>5s} profiles {sys.argv[2]:4s} lines {j["kernel_ms_per_step"]["lines"]:.4f} step {j["ms_per_step"]:.4f}')
P
done; done
