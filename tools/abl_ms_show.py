#!/usr/bin/env python3
"""Per-workgroup counters of the ablated runs of tools/abl_ms_pmc.sh: python tools/abl_ms_show.py 0 2 1"""
import json, sys
keys = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_VALU_FMA_F64",
        "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_TRANS_F64", "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64", "SQ_INSTS_VALU_CVT",
        "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE"]
for a in sys.argv[1:]:
    c = json.load(open(f"gpurun_out/ab/pmc_abl{a}.json"))["per_launch"]["c4"]["lines_kernel"]
    d = json.load(open(f"gpurun_out/ab/d_abl{a}.json"))
    nwg = 10944.0   # configs[3] whole: 171 groups of six profiles x 64 layers
    print("ablate", a, f'{d["kernel_ms_per_step"]["lines"]:.4f} ms', " ".join(f'{k.replace("SQ_INSTS_", "").replace("SQ_", "")}={c.get(k, 0) / nwg:.5g}' for k in keys))
