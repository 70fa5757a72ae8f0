#!/bin/bash
# Instruction counts of lines_kernel by stage for one bench workload: SQ_INSTS_* of the prologue-only (abl_LOOP), prologue +
# prepare (abl_EVAL) and full builds (tools/build_variant.sh).  GPU box, repo root; counters only, no trace domains.
# usage: tools/pmc_ablation.sh WORKLOAD [PROFILES]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
export PMC_WORKLOAD=$1 PMC_PROFILES=${2:-128}
cat > gpurun_out/_steps.py <<'PY'
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch, bench
res = bench.Resident(os.environ["PMC_WORKLOAD"], 0, 0, int(os.environ.get("PMC_PROFILES", "128")))
for _ in range(3): res.batch.step()
torch.cuda.synchronize()
PY
for V in LOOP EVAL FULL; do
  if [ $V = FULL ]; then unset MONORTM_HIP_LIB; else export MONORTM_HIP_LIB=$GRAFT_REPO_ROOT/build_dbg/libmonortm_hip_abl_$V.so; fi
  rm -rf gpurun_out/abl_$V
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVES \
    --output-format csv -d gpurun_out/abl_$V -- python3 gpurun_out/_steps.py > gpurun_out/abl_$V.log 2>&1
  python3 - gpurun_out/abl_$V "$1 $V" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:40]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(k, r["Counter_Name"])] += 1
for k, cs in acc.items():
    if "lines_kernel" in k:
        w = cs["SQ_WAVES"] / max(n[(k, "SQ_WAVES")], 1)
        print(sys.argv[2], k[-30:], "waves", int(w), {c: f"{v / max(n[(k, c)], 1) / w:.1f}" for c, v in cs.items() if c != "SQ_WAVES"}, "per wave")
PY
done
