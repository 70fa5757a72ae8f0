cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
python3 -m pytest tests/test_hip_parity.py tests/test_fullsize.py tests/test_fuzz_gpu.py -m gpu -x -q > gpurun_out/g_tests.txt 2>&1; tail -3 gpurun_out/g_tests.txt
for W in c4brd c4; do tools/ab_libs.sh $W wn - ; done > gpurun_out/g_ab.txt 2>&1; cat gpurun_out/g_ab.txt
