#!/bin/bash
# lines-kernel time of one bench workload under several settings of ONE environment variable (GPU box, repo root).
# usage: tools/ab_env.sh WORKLOAD VAR value1 value2 ...   ("-" = unset)     e.g. tools/ab_env.sh c4shard MONORTM_STAGGER - 1 2 4
cd "$GRAFT_REPO_ROOT" || exit 1
W=$1; VAR=$2; shift 2
for V in "$@"; do
  if [ "$V" = "-" ]; then unset $VAR; else export $VAR=$V; fi
  timeout -k 10 240 python bench.py --workload $W --no-extra --no-pmc --no-cpu-baseline --steps ${STEPS:-100} --warmup 10 > gpurun_out/abe_${W}_$V.json 2> gpurun_out/abe_${W}_$V.err || { echo "FAILED $W $V"; tail -5 gpurun_out/abe_${W}_$V.err; exit 1; }
  python - gpurun_out/abe_${W}_$V.json $W "$VAR=$V" <<'PY'
import json, sys
b = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], sys.argv[3], "ms/step %.4f" % b["ms_per_step"], "kernels", {k: round(v, 4) for k, v in b["kernel_ms_per_step"].items()})
PY
done
