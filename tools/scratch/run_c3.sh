cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
PMC_SETS="FETCH_SIZE;WRITE_SIZE" tools/pmc_libs.sh c3 - > gpurun_out/c3_xcd_pmc.txt 2>&1 ; cat gpurun_out/c3_xcd_pmc.txt
