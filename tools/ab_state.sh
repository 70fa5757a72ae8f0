#!/bin/bash
# A/B of the two line-sum kernels on the GPU box (repo root): lines-kernel time of one bench workload each way.
# usage: tools/ab_state.sh "c4full c4shard c5full"
cd "$GRAFT_REPO_ROOT" || exit 1
for W in ${1:-c4full}; do
  for V in wn state; do
    MONORTM_LINES_KERNEL=$V timeout -k 10 240 python bench.py --workload $W --no-extra --no-pmc --no-cpu-baseline --steps 30 --warmup 5 > gpurun_out/ab_${W}_$V.json 2> gpurun_out/ab_${W}_$V.err || { echo "FAILED $W $V"; tail -5 gpurun_out/ab_${W}_$V.err; exit 1; }
    python - gpurun_out/ab_${W}_$V.json $W $V <<'PY'
import json, sys
b = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], sys.argv[3], "ms/step %.4f" % b["ms_per_step"], "kernels", {k: round(v, 4) for k, v in b["kernel_ms_per_step"].items()}, "value %.3e" % b["value"])
PY
  done
done
