cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
python3 -m pytest tests -m gpu -x -q > gpurun_out/mw_tests.txt 2>&1; tail -5 gpurun_out/mw_tests.txt
for W in c4shard c4 c5full; do python3 bench.py --workload $W --no-extra --no-pmc --no-cpu-baseline > gpurun_out/mw_$W.json 2> gpurun_out/mw_$W.err && python3 -c "
import json; j=json.load(open('gpurun_out/mw_$W.json')); print('$W', j['value'], j['ms_per_step'], j['kernel_ms_per_step'])"; done
