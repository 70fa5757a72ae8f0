#!/bin/bash
# Per-kernel average durations of one bench workload (rocprofv3 --kernel-trace --stats; GPU box, repo root).
# usage: tools/trace_kernels.sh WORKLOAD [TAG]     -> gpurun_out/TAG_WORKLOAD_kernel_stats.txt
W=$1; TAG=${2:-trace}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/${TAG}_trace_$W
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_trace_$W -- \
  python3 bench.py --workload $W --no-extra --no-pmc --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/${TAG}_trace_$W.log 2>&1 || { tail -5 gpurun_out/${TAG}_trace_$W.log; exit 1; }
python3 - gpurun_out/${TAG}_trace_$W > gpurun_out/${TAG}_${W}_kernel_stats.txt <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"].replace("void (anonymous namespace)::", "")[:44]
        if float(r["TotalDurationNs"]) > 2e5:
            print(f"{n:46s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs']) / 1e3:9.1f}  min {float(r['MinNs']) / 1e3:8.1f} max {float(r['MaxNs']) / 1e3:8.1f}")
PY
cat gpurun_out/${TAG}_${W}_kernel_stats.txt
