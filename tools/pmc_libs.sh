#!/bin/bash
# Counters of the line-sum kernel for one bench workload with several builds of the library (GPU box, repo root).
# usage: tools/pmc_libs.sh WORKLOAD lib1.so lib2.so ...   ("-" = the shipped build; counters only, no trace domains)
# PMC_SETS="A B;C D" overrides the counter sets (one rocprofv3 pass each); PMC_MATCH="lines_,far_" the kernels reported.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
W=$1; shift
cat > gpurun_out/_steps.py <<'PY'
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch, bench
res = bench.Resident(os.environ["PMC_WORKLOAD"], 0, 0, int(os.environ.get("PMC_PROFILES", "128")))
for _ in range(3): res.batch.step()
torch.cuda.synchronize()
PY
export PMC_WORKLOAD=$W
for LIB in "$@"; do
  if [ "$LIB" = "-" ]; then unset MONORTM_HIP_LIB; else export MONORTM_HIP_LIB=$PWD/$LIB; fi
  T=$(basename "$LIB" .so)
  IFS=';' read -ra SETS <<< "${PMC_SETS:-SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES;SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_ACTIVE_INST_ANY;SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE}"
  for SET in "${SETS[@]}"; do
    rm -rf gpurun_out/pmcl_$T
    timeout -k 10 200 rocprofv3 --pmc $SET --output-format csv -d gpurun_out/pmcl_$T -- python3 gpurun_out/_steps.py > gpurun_out/pmcl_$T.log 2>&1
    python3 - gpurun_out/pmcl_$T "$W $T" <<'PY'
import csv, glob, os, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void (anonymous namespace)::", "")[:40]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(k, r["Counter_Name"])] += 1
for k, cs in acc.items():
    if any(t in k for t in os.environ.get("PMC_MATCH", "lines_").split(",")):
        print(sys.argv[2], k[:40], {c: f"{v / max(n[(k, c)], 1):.5g}" for c, v in cs.items()}, "per launch")
PY
  done
done
