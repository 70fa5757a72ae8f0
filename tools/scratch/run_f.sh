cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
python3 -m pytest tests/test_hip_parity.py tests/test_fullsize.py tests/test_fuzz_gpu.py -m gpu -x -q > gpurun_out/f_tests.txt 2>&1; tail -4 gpurun_out/f_tests.txt
for W in c5full c5 c4; do python3 bench.py --workload $W --no-extra --no-pmc --no-cpu-baseline > gpurun_out/f_$W.json 2> gpurun_out/f_$W.err && python3 -c "
import json; j=json.load(open('gpurun_out/f_$W.json')); print('$W', j['value'], j['ms_per_step'], j['kernel_ms_per_step'])"; done
