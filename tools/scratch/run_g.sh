cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
python3 -m pytest tests/test_hip_parity.py tests/test_fullsize.py tests/test_fuzz_gpu.py -m gpu -x -q > gpurun_out/g_tests.txt 2>&1; tail -3 gpurun_out/g_tests.txt
tools/ab_libs.sh c4 wn - > gpurun_out/g_ab.txt 2>&1; tools/ab_libs.sh c4shard wn - >> gpurun_out/g_ab.txt 2>&1; tools/ab_libs.sh c2lc wn - >> gpurun_out/g_ab.txt 2>&1; tools/ab_libs.sh c5full wn - >> gpurun_out/g_ab.txt 2>&1;  tools/ab_libs.sh c3 wn - >> gpurun_out/g_ab.txt 2>&1; cat gpurun_out/g_ab.txt
tools/pmc_ablation.sh c4 1024 > gpurun_out/abl_c4.txt 2>&1; cat gpurun_out/abl_c4.txt
