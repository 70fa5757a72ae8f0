cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
python3 -m pytest tests/test_fortran_dropin.py tests/test_reference_driver_dropin.py tests/test_xsec_driver.py tests/test_abi.py tests/test_hip_parity.py -m gpu -x -q > gpurun_out/x_tests.txt 2>&1; tail -4 gpurun_out/x_tests.txt
python3 tools/fuzz_more.py 70000 400 > gpurun_out/r04_fuzz_more_400seeds.log 2>&1; tail -8 gpurun_out/r04_fuzz_more_400seeds.log
python3 tools/fuzz_more.py 90000 100 allmol > gpurun_out/r04_fuzz_more_allmol_100seeds.log 2>&1; tail -8 gpurun_out/r04_fuzz_more_allmol_100seeds.log
