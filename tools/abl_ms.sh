#!/bin/bash
# stage pricing of lines_ms_kernel on one box: tools/abl_ms.sh LIB "A1 A2 ..." [WORKLOAD] (LIB = an experiment build, MONORTM_MS_ABLATE values)
mkdir -p gpurun_out/ab
export MONORTM_HIP_LIB=$PWD/$1
W=${3:-c4}
for r in 1 2; do for a in $2; do
  MONORTM_MS_ABLATE=$a python bench.py --workload $W --steps 100 --no-extra --no-cpu-baseline --no-pmc --detail-file gpurun_out/ab/d.json 2>gpurun_out/ab/err.txt >/dev/null || { tail -2 gpurun_out/ab/err.txt; continue; }
  python - $a $1 <<'P'
import json,sys
j=json.load(open("gpurun_out/ab/d.json")); print(f'{sys.argv[2]} ablate {sys.argv[1]} lines {j["kernel_ms_per_step"]["lines"]:.4f}')
P
done; done
