#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/ic
for k in wn ms; do
  rm -rf gpurun_out/ic/$k
  MONORTM_LINES_KERNEL=$k rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d gpurun_out/ic/$k -- python3 bench.py --pmc-child --pmc-workloads c4 --pmc-manifest gpurun_out/ic/man_$k.json --profiles-per-gpu 128 > gpurun_out/ic/log_$k.txt 2>&1
  python3 - gpurun_out/ic/$k <<'P'
import csv,glob,sys,collections
tot=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name=r["Kernel_Name"]
        if "lines" in name and "kernel" in name:
            key="ms" if "lines_ms" in name else "wn"
            tot[key][r["Counter_Name"]]+=float(r["Counter_Value"])
for k,v in tot.items(): print(k, {a:round(b/3e6,2) for a,b in v.items()})
P
done
