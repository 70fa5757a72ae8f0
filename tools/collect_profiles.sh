#!/bin/bash
# Collect the measurement artefacts of a round on the GPU box (run through gpurun from the repo root):
#   tools/collect_profiles.sh TAG     e.g. TAG=r01_c
# 1. rocprofv3 --kernel-trace --stats of the default bench.py command  -> gpurun_out/TAG_c4shard_kernel_stats.csv
# 2. PMC passes (separate runs, no trace domains): FETCH_SIZE, WRITE_SIZE, SQ counters -> gpurun_out/TAG_c4shard_pmc_summary.csv
# 3. the bench lines themselves (c4shard default, c3, c5)                -> gpurun_out/TAG_*_bench.json
# The summaries are copied into profiles/ by hand afterwards (profiles/ is tracked, gpurun_out/ is scratch).
set -o pipefail
TAG=${1:-r01}
OUT=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p $OUT
BENCH="python3 bench.py --no-cpu-baseline --no-single"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -- $BENCH > $OUT/${TAG}_trace.log 2>&1 || exit 1
cp $(ls $OUT/${TAG}_trace/*/*kernel_stats.csv | head -1) $OUT/${TAG}_c4shard_kernel_stats.csv
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $OUT/${TAG}_pmc_$C -- $BENCH --steps 5 > $OUT/${TAG}_pmc_$C.log 2>&1 || exit 1
done
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE \
  --output-format csv -d $OUT/${TAG}_pmc_sq1 -- $BENCH --steps 5 > $OUT/${TAG}_pmc_sq1.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_TRANS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_ACTIVE_INST_SCA \
  --output-format csv -d $OUT/${TAG}_pmc_sq2 -- $BENCH --steps 5 > $OUT/${TAG}_pmc_sq2.log 2>&1 || exit 1
python3 tools/pmc_report.py $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE $OUT/${TAG}_pmc_sq1 $OUT/${TAG}_pmc_sq2 --kernel _kernel \
  --csv $OUT/${TAG}_c4shard_pmc_summary.csv || exit 1
timeout -k 10 400 python3 bench.py > $OUT/${TAG}_c4shard_bench.json 2> $OUT/${TAG}_c4shard_bench.err || exit 1
timeout -k 10 300 python3 bench.py --workload c3 > $OUT/${TAG}_c3_bench.json 2> $OUT/${TAG}_c3_bench.err || exit 1
timeout -k 10 300 python3 bench.py --workload c5 > $OUT/${TAG}_c5_bench.json 2> $OUT/${TAG}_c5_bench.err || exit 1
head -4 $OUT/${TAG}_c4shard_kernel_stats.csv
grep -E "FETCH_SIZE|WRITE_SIZE" $OUT/${TAG}_c4shard_pmc_summary.csv | grep lines_kernel
cut -c1-400 $OUT/${TAG}_c4shard_bench.json
