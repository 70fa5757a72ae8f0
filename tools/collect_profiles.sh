#!/bin/bash
# Collect the measurement artefacts of a round on the GPU box (run through gpurun from the repo root):
#   tools/collect_profiles.sh TAG     e.g. TAG=r02_a
# 1. the default bench.py command (headline + c3 / c5 / c2lc / single profile + drop-in latency + CPU baseline), with
#    its live counter collection saved   -> gpurun_out/TAG_bench.json (compact line), TAG_bench_detail.json, TAG_pmc_per_launch.json
# 2. per workload: rocprofv3 --kernel-trace --stats of `bench.py --workload W --no-extra --no-pmc --no-cpu-baseline`
#    (the average duration of lines_kernel must agree with roofline.avg_launch_ms of the same workload in 1.)
#                                                                 -> gpurun_out/TAG_W_kernel_stats.csv
# 3. a readable per-kernel counter table from the saved counters -> gpurun_out/TAG_pmc_summary.csv
# The summaries are copied into profiles/ by hand afterwards (profiles/ is tracked, gpurun_out/ is scratch).
# rocprofv3: program directly after `--`, counters and traces never in the same run.
set -o pipefail
TAG=${1:-r04}
OUT=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p $OUT
# (stdout = the compact line the driver parses; the full record goes to the detail file)
timeout -k 10 700 python3 bench.py --save-pmc $OUT/${TAG}_pmc_per_launch.json --detail-file $OUT/${TAG}_bench_detail.json > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err || exit 1
for W in c4 c4shard c3 c5 c5full c2lc c4brd c2real; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace_$W -- \
    python3 bench.py --workload $W --no-extra --no-pmc --no-cpu-baseline > $OUT/${TAG}_trace_$W.log 2>&1 || exit 1
  python3 - "$OUT/${TAG}_trace_$W" "$OUT/${TAG}_${W}_kernel_stats.csv" <<'EOF' || exit 1
import csv, glob, sys
rows = {}
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):   # one file per process: merge
    for r in csv.DictReader(open(f)):
        a = rows.setdefault(r["Name"], [0, 0, 1e30, 0])
        a[0] += int(r["Calls"]); a[1] += int(r["TotalDurationNs"]); a[2] = min(a[2], int(r["MinNs"])); a[3] = max(a[3], int(r["MaxNs"]))
w = csv.writer(open(sys.argv[2], "w", newline=""))
w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs"])
for n, a in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    w.writerow([n, a[0], a[1], f"{a[1] / a[0]:.1f}", a[2], a[3]])
EOF
done
python3 - "$OUT/${TAG}_pmc_per_launch.json" "$OUT/${TAG}_pmc_summary.csv" <<'EOF' || exit 1
import csv, json, sys
j = json.load(open(sys.argv[1]))
w = csv.writer(open(sys.argv[2], "w", newline=""))
w.writerow(["csrc_hash", j["csrc_hash"], "profiles_per_gpu", j["profiles_per_gpu"]])
w.writerow(["workload", "kernel", "counter", "mean_per_launch"])
for wl, ks in j["per_launch"].items():
    for k, cs in ks.items():
        for c, v in sorted(cs.items()):
            w.writerow([wl, k, c, f"{v:.6g}" if isinstance(v, (int, float)) else " ".join(map(str, v))])
EOF
python3 - "$OUT/${TAG}_bench_detail.json" <<'EOF'
import json, sys
j = json.load(open(sys.argv[1]))
r = j["roofline"]
print("headline", f'{j["value"]:.4g}', "evals/s", j["kernel_ms_per_step"], "fp64 frac", r["frac"], "valu busy", r.get("valu_busy"), "|", r["counter_source"])
for k, v in j.get("workloads", {}).items():
    print(k, f'{v["value"]:.4g}', v["kernel_ms_per_step"], "frac", v.get("roofline", {}).get("frac"))
print("dropin", j.get("dropin", {}).get("ms_per_profile"), "cpu", j.get("cpu_baseline", {}).get("value"))
EOF
for W in c4 c4shard c3 c5 c5full c2lc c4brd c2real; do head -2 $OUT/${TAG}_${W}_kernel_stats.csv | tail -1 | cut -c1-160; done
