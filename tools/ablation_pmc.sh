#!/bin/bash
# Instruction counts per wave of lines_kernel by stage: SQ_INSTS_* of the prologue-only, prologue + prepare and full builds
# (tools/debug_builds.sh) on the c4shard batch.  Run on the GPU box from the repo root; counters only, no trace domains.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
cat > gpurun_out/_steps.py <<'PY'
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch, bench
res = bench.Resident("c4shard", 0, 0, 128)
for _ in range(3): res.batch.step()
torch.cuda.synchronize()
PY
for V in LOOP EVAL FULL; do
  if [ $V = FULL ]; then unset MONORTM_HIP_LIB; else export MONORTM_HIP_LIB=$GRAFT_REPO_ROOT/build_dbg/libmonortm_hip_abl_$V.so; fi
  rm -rf gpurun_out/abl_$V
  timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES \
    --output-format csv -d gpurun_out/abl_$V -- python3 gpurun_out/_steps.py > gpurun_out/abl_$V.log 2>&1
  python3 - gpurun_out/abl_$V $V <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:40]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_INSTS_VALU": n[k] += 1
for k, cs in acc.items():
    if "lines_kernel" in k:
        print(sys.argv[2], {c: f"{v / max(n[k], 1) / 8192:.1f}" for c, v in cs.items()}, "per wave (8192 one-wave workgroups)")
PY
done
