#!/bin/bash
# lines-kernel time of the c4 shape against the number of profiles, for the settings of one environment variable (GPU box).
# usage: tools/sweep_profiles.sh VAR "v1 v2" "32 64 128 ..."
cd "$GRAFT_REPO_ROOT" || exit 1
VAR=$1
for P in $3; do
  for V in $2; do
    export $VAR=$V
    python bench.py --workload c4shard --profiles-per-gpu $P --no-extra --no-pmc --no-cpu-baseline --steps 60 --warmup 10 2>/dev/null > gpurun_out/_sw.json
    python - $P "$VAR=$V" <<'PY'
import json, sys
b = json.loads(open("gpurun_out/_sw.json").read().strip().splitlines()[-1])
print(sys.argv[1], sys.argv[2], "lines %.4f ms" % b["kernel_ms_per_step"]["lines"], "step %.4f ms" % b["ms_per_step"], "value %.3e" % b["value"])
PY
  done
done
