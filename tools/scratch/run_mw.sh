cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for W in c4shard c4 c5full; do python3 bench.py --workload $W --no-extra --no-pmc --no-cpu-baseline > gpurun_out/mw_$W.json 2> gpurun_out/mw_$W.err && python3 -c "
import json; j=json.load(open('gpurun_out/mw_$W.json')); print('$W', j['value'], j['ms_per_step'], j['kernel_ms_per_step'])"; done
