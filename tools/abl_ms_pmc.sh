#!/bin/bash
# counters of lines_ms_kernel's stages on one box: tools/abl_ms_pmc.sh LIB "A1 A2 ..." (experiment build; MONORTM_MS_ABLATE values)
mkdir -p gpurun_out/ab
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
export MONORTM_HIP_LIB=$PWD/$1
for a in $2; do
  MONORTM_MS_ABLATE=$a python3 bench.py --workload c4 --steps 20 --no-extra --no-cpu-baseline --save-pmc gpurun_out/ab/pmc_abl$a.json --detail-file gpurun_out/ab/d_abl$a.json >/dev/null 2>gpurun_out/ab/err_abl$a.txt || { tail -3 gpurun_out/ab/err_abl$a.txt; continue; }
  python3 - $a <<'P'
import json,sys
j=json.load(open(f"gpurun_out/ab/pmc_abl{sys.argv[1]}.json"))
c=j["per_launch"]["c4"]["lines"]
keys=["SQ_INSTS_VALU","SQ_INSTS_SALU","SQ_INSTS_LDS","SQ_INSTS_SMEM","SQ_INSTS_VMEM_RD","SQ_INSTS_VMEM_WR","SQ_INSTS_VALU_FMA_F64","SQ_INSTS_VALU_MUL_F64","SQ_INSTS_VALU_ADD_F64","SQ_INSTS_VALU_TRANS_F64","SQ_INSTS_VALU_INT32","SQ_INSTS_VALU_INT64","SQ_INSTS_VALU_CVT","SQ_ACTIVE_INST_VALU","SQ_WAIT_INST_ANY","SQ_WAVE_CYCLES","GRBM_GUI_ACTIVE"]
print("ablate",sys.argv[1]," ".join(f"{k.replace('SQ_INSTS_','').replace('SQ_','')}={c.get(k,0):.4g}" for k in keys))
P
done
