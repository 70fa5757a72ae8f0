#!/bin/bash
# lines-kernel time of bench workloads with several builds of the library (GPU box, repo root).
# usage: tools/ab_libs.sh "W1 W2 .." lib1.so lib2.so ...   (paths relative to the repo root; "-" = the shipped build)
cd "$GRAFT_REPO_ROOT" || exit 1
WS=$1; shift
for W in $WS; do
for LIB in "$@"; do
  if [ "$LIB" = "-" ]; then unset MONORTM_HIP_LIB; else export MONORTM_HIP_LIB=$PWD/$LIB; fi
  T=$(basename "$LIB" .so)
  timeout -k 10 240 python bench.py --workload $W --no-extra --no-pmc --no-cpu-baseline --steps 40 --warmup 5 > gpurun_out/abl_${W}_$T.json 2> gpurun_out/abl_${W}_$T.err || { echo "FAILED $W $T"; tail -5 gpurun_out/abl_${W}_$T.err; continue; }
  python - gpurun_out/abl_${W}_$T.json $W $T <<'PY'
import json, sys
b = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], sys.argv[3], "ms/step %.4f" % b["ms_per_step"], "kernels", {k: round(v, 4) for k, v in b["kernel_ms_per_step"].items()}, flush=True)
PY
done
done
