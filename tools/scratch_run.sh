timeout -k 10 600 python -m pytest tests/test_hip_parity.py tests/test_fuzz_gpu.py -m gpu -x -q -k "far or dense or physics or infrared" 2>&1 | grep -v amdgpu | tail -2
tools/trace_kernels.sh c3 r05_y | grep "far_kernel\|lines_kernel"; tools/ab_libs.sh c3 - | tail -1
