#!/usr/bin/env python3
"""Opcode census of the gfx950 code of lines_kernel: which instructions make up the part of the vector
issue that is NOT FP64 arithmetic, per stage of the kernel and per inner loop.

    python tools/isa_census.py [--kernel d11] [--out profiles/r03_isa_census] [--classes gpurun_out/classes_c4shard.json]

1. compiles monortm_amd/csrc/lines_kernel.hip for gfx950 to assembly with line tables (`-S -gline-tables-only`: the code is
   the one of the shipped build, plus `.loc` directives);
2. splits the chosen instantiation (d11 = lines_kernel<double,1,1,false>, the c4shard kernel; d42 = <double,4,2,false>, c3;
   f12 / f22 = the float ones of c5) into basic blocks, finds the innermost loops (back edges) and attributes every instruction
   to a STAGE through the source line it was generated from (the inlined device function of lines_device.hpp / lineshape.hpp,
   or the region of lines_kernel.hip); helpers that are inlined everywhere (frcp, exp_prep, dpp_move, ...) take the stage of
   the surrounding instructions;
3. histograms opcode groups per stage (static) and per innermost loop (per trip), and - given trip counts - weights the
   loops to a dynamic estimate per wave.

Outputs <out>_<kernel>_stages.csv, <out>_<kernel>_loops.csv and a text summary on stdout.
"""
from __future__ import annotations

import argparse
import collections
import csv
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "monortm_amd", "csrc")
KERNELS = {"d11": "lines_kernelIdLi1ELi1ELb0E", "d12": "lines_kernelIdLi1ELi2ELb0E", "d22": "lines_kernelIdLi2ELi2ELb0E",
           "d42": "lines_kernelIdLi4ELi2ELb0E", "f11": "lines_kernelIfLi1ELi1ELb0E", "f12": "lines_kernelIfLi1ELi2ELb0E",
           "f22": "lines_kernelIfLi2ELi2ELb0E", "f42": "lines_kernelIfLi4ELi2ELb0E", "d11b": "lines_kernelIdLi1ELi1ELb1E", "f14": "lines_kernelIfLi1ELi4ELb0E"}

GROUPS = [  # first match wins
    ("fp64_fma", r"v_fma_f64|v_fmac_f64"), ("fp64_mul", r"v_mul_f64"), ("fp64_add", r"v_add_f64"),
    ("fp64_trans", r"v_(rcp|rsq|sqrt)_f64"), ("fp64_cmp", r"v_cmpx?_\w+_f64|v_cmp_class_f64"),
    ("fp64_other", r"v_(max|min|ldexp|frexp_\w+|fract|trunc|floor|ceil|rndne|div_\w+)_f64|v_cvt_\w*f64\w*"),
    ("fp32_pk", r"v_pk_(fma|mul|add)_f32"), ("fp32_arith", r"v_(fma|fmac|mul|add|sub|subrev|mac|mad)_f32|v_fmaak_f32|v_fmamk_f32"),
    ("fp32_trans", r"v_(rcp|rsq|sqrt|exp|log|sin|cos)_f32"), ("fp32_cmp", r"v_cmpx?_\w+_f32"),
    ("fp32_other", r"v_\w+_f32|v_cvt_\w+"),
    ("cndmask", r"v_cndmask_b32"), ("mov", r"v_mov_b32|v_mov_b64|v_accvgpr_\w+|v_pk_mov_b32"),
    ("lane_xfer", r"v_readlane_b32|v_readfirstlane_b32|v_writelane_b32|v_permlane\w+|v_mov_b32_dpp|v_\w+_dpp|ds_bpermute_b32|ds_swizzle_b32|v_mbcnt\w+"),
    ("int_cmp", r"v_cmpx?_\w+_[iu](32|64|16)"), ("int_valu", r"v_\w+"),
    ("lds", r"ds_\w+"), ("vmem", r"(global|buffer|flat|scratch)_\w+"),
    ("smem", r"s_load_\w+|s_buffer_load_\w+|s_memtime|s_memrealtime"), ("waitcnt", r"s_waitcnt\w*|s_nop"),
    ("branch", r"s_cbranch_\w+|s_branch|s_endpgm|s_setpc_b64|s_swappc_b64|s_barrier"), ("salu", r"s_\w+"),
]
GROUPS = [(g, re.compile(r"^(?:" + p + r")(?:_e32|_e64|_sdwa|_dpp)?$")) for g, p in GROUPS]
VALU_GROUPS = {"fp64_fma", "fp64_mul", "fp64_add", "fp64_trans", "fp64_cmp", "fp64_other", "fp32_pk", "fp32_arith", "fp32_trans",
               "fp32_cmp", "fp32_other", "cndmask", "mov", "lane_xfer", "int_cmp", "int_valu"}
FP64_ARITH = {"fp64_fma", "fp64_mul", "fp64_add", "fp64_trans"}      # what SQ_INSTS_VALU_{FMA,MUL,ADD,TRANS}_F64 count
HELPERS = {"frcp", "frcp_any", "exp_prep", "dpp_move", "wave_sum", "wave_min", "widen", "uni64", "pk_fma", "splat", "swap_add32",
           "swap_add16", "row_sum16", "rp", "wp", "tips_atob", "open_runs8", "close_runs8", "far_moment_of_lane"}


def group_of(op: str) -> str:
    if op.endswith("_dpp"):
        return "lane_xfer"
    for g, rx in GROUPS:
        if rx.match(op):
            return g
    return "other"


def source_functions(path: str):
    """[(first line, name)] of the device functions / kernels of one source file (definitions start at column 0 or after
    `template <...>`; good enough for this code base's layout)."""
    out = []
    rx = re.compile(r"^(?:__host__ )?(?:__device__|__global__)[^;(]*?\b([A-Za-z_]\w*)\s*\(")
    rx2 = re.compile(r"^(?:static |inline )*(?:void|int|double|float|bool|cx|LinePhys|f2|HotA)\s+([A-Za-z_]\w*)\s*\(")
    for i, ln in enumerate(open(path), 1):
        m = rx.match(ln) or rx2.match(ln)
        if m and not ln.rstrip().endswith(";"):
            out.append((i, m.group(1)))
    return out


def kernel_regions(path: str):
    """Stage boundaries inside lines_kernel.hip by marker comments."""
    marks = []
    for i, ln in enumerate(open(path), 1):
        if "// ================= prepare" in ln:
            marks.append((i, "k:prepare"))
        elif "// ================= evaluate" in ln:
            marks.append((i, "k:evaluate_glue"))
        elif "for (int base = vbeg" in ln:
            marks.append((i, "k:chunk_loop"))
        elif ln.startswith("template <typename R, bool IBRD>") and not marks_has(marks, "k:physics_kernel"):
            marks.append((i, "k:physics_kernel"))
    return marks


def marks_has(marks, name):
    return any(n == name for _, n in marks)


class StageMap:
    def __init__(self):
        self.files = {}
        for f in ("lines_device.hpp", "lineshape.hpp", "lines_kernel.hip", "device_common.hpp"):
            p = os.path.join(CSRC, f)
            fn = source_functions(p)
            if f == "lines_kernel.hip":
                fn = [(1, "k:prologue")] + [(ln, nm) for ln, nm in kernel_regions(p)]
                fn.sort()
            self.files[f] = fn

    def stage(self, fname: str, line: int) -> str:
        base = os.path.basename(fname)
        fn = self.files.get(base)
        if fn is None:
            return "lib:" + base
        cur = "?"
        for ln, nm in fn:
            if ln <= line:
                cur = nm
            else:
                break
        return cur


def compile_asm(out_s: str, extra=()):
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-Wno-pass-failed",
           "-Wno-unused-const-variable", "-Wno-unused-command-line-argument", "--cuda-device-only", "-S", "-gline-tables-only", *extra,
           os.path.join(CSRC, "lines_kernel.hip"), "-o", out_s]
    subprocess.check_call(cmd)


def parse_kernel(asm_path: str, mangled_part: str, smap: StageMap):
    files = {}
    insts = []   # (block label, opcode, operands, stage_raw)
    labels = {}  # label -> index of first instruction
    in_k = False
    cur_loc = ("?", 0)
    block = "entry"
    rx_file = re.compile(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?')
    rx_loc = re.compile(r"\s*\.loc\s+(\d+)\s+(\d+)")
    rx_lab = re.compile(r"^(\.LBB\d+_\d+):")
    rx_ins = re.compile(r"^\t([a-z_0-9]+)(?:\s+(.*?))?(?:\s*;.*)?$")
    rx_num = re.compile(r"^\s*(\d+):\s*$")
    in_asm, asm_id = False, 0
    numeric = collections.defaultdict(list)   # (asm block, label number) -> [instruction index] of a hand-written loop label
    for ln in open(asm_path):
        if in_k and ";;#ASMSTART" in ln:
            in_asm, asm_id = True, asm_id + 1
            continue
        if in_k and ";;#ASMEND" in ln:
            in_asm = False
            continue
        if in_k and in_asm:
            m = rx_num.match(ln)
            if m:   # numeric local label of lines_asm.hpp: a basic block of its own, unique per occurrence
                block = f"A{asm_id}_{m.group(1)}"
                while block in labels:
                    block += "'"
                labels[block] = len(insts)
                numeric[(asm_id, m.group(1))].append((len(insts), block))
                continue
            m = rx_ins.match(ln)
            if m and not m.group(1).startswith("."):
                insts.append((block, m.group(1), (m.group(2) or "") + f" @asm{asm_id}", "asm_loops"))
            continue
        m = rx_file.match(ln)
        if m:
            files[int(m.group(1))] = m.group(3) or m.group(2)
            continue
        if not in_k:
            if ln.startswith("_Z") and mangled_part in ln.split(":")[0] and ln.rstrip().split(";")[0].rstrip().endswith(":"):
                in_k = True
            continue
        if ln.startswith(".Lfunc_end") or ln.startswith("\t.section\t.rodata") or ln.startswith("\t.amdhsa_kernel"):
            break
        m = rx_loc.match(ln)
        if m:
            cur_loc = (files.get(int(m.group(1)), "?"), int(m.group(2)))
            continue
        m = rx_lab.match(ln)
        if m:
            block = m.group(1)
            labels[block] = len(insts)
            continue
        m = rx_ins.match(ln)
        if m and not m.group(1).startswith("."):
            op = m.group(1)
            if op in ("s_code_end",):
                continue
            insts.append((block, op, m.group(2) or "", smap.stage(*cur_loc)))
    # resolve `NNf` / `NNb` branch targets of the hand-written loops to the unique block names
    rx_tgt = re.compile(r"^(\d+)([fb]) @asm(\d+)$")
    for i, (blk, op, args, st) in enumerate(insts):
        if args.endswith(tuple(f"@asm{k}" for k in range(1, asm_id + 1))):
            m = rx_tgt.match(args.strip())
            if m and (op.startswith("s_cbranch") or op == "s_branch"):
                occ = numeric.get((int(m.group(3)), m.group(1)), [])
                if m.group(2) == "f":
                    cand = [b for pos, b in occ if pos > i]
                    tgt = cand[0] if cand else ""
                else:
                    cand = [b for pos, b in occ if pos <= i]
                    tgt = cand[-1] if cand else ""
                insts[i] = (blk, op, tgt, st)
            else:
                insts[i] = (blk, op, args.rsplit(" @asm", 1)[0], st)
    return insts, labels


def resolve_helpers(insts):
    """helpers inlined everywhere take the stage of the nearest non-helper instruction of the same basic block (else of the
    neighbouring blocks)"""
    stages = [s for _, _, _, s in insts]
    n = len(insts)
    fixed = list(stages)
    for i, s in enumerate(stages):
        if s in HELPERS or s == "?" or s.startswith("lib:"):
            best = None
            for d in range(1, n):
                for j in (i - d, i + d):
                    if 0 <= j < n and stages[j] not in HELPERS and stages[j] != "?" and not stages[j].startswith("lib:"):
                        best = stages[j]
                        break
                if best:
                    break
            fixed[i] = (best or "?")
    return fixed


def find_loops(insts, labels):
    """innermost loops = back edges whose body holds no other back edge"""
    back = []
    for i, (_, op, args, _) in enumerate(insts):
        if op.startswith("s_cbranch") or op == "s_branch":
            tgt = args.strip().split()[-1] if args.strip() else ""
            if tgt in labels and labels[tgt] <= i:
                back.append((labels[tgt], i))
    inner = [(a, b) for a, b in back if not any((a2 >= a and b2 <= b) and (a2, b2) != (a, b) for a2, b2 in back)]
    return sorted(set(inner)), sorted(set(back))


def histogram(insts, stages, lo, hi):
    h = collections.Counter()
    for i in range(lo, hi + 1):
        h[group_of(insts[i][1])] += 1
    return h


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="d11", choices=sorted(KERNELS))
    ap.add_argument("--asm", default="/tmp/isa/lines_g.s")
    ap.add_argument("--recompile", action="store_true")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r03_isa_census"))
    ap.add_argument("--trips", default=None, help="json: {loop id: trips per wave} for the weighted estimate")
    a = ap.parse_args()
    if a.recompile or not os.path.exists(a.asm):
        os.makedirs(os.path.dirname(a.asm), exist_ok=True)
        compile_asm(a.asm)
    smap = StageMap()
    insts, labels = parse_kernel(a.asm, KERNELS[a.kernel], smap)
    if not insts:
        sys.exit("kernel not found in " + a.asm)
    stages = resolve_helpers(insts)
    inner, back = find_loops(insts, labels)
    allg = [g for g, _ in GROUPS] + ["other"]
    # ---- static census per stage
    per_stage = collections.defaultdict(collections.Counter)
    for (blk, op, args, _), st in zip(insts, stages):
        per_stage[st][group_of(op)] += 1
    with open(f"{a.out}_{a.kernel}_stages.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["stage", "instructions", "valu", "fp64_arith", "fp64_arith_share_of_valu", *allg])
        for st, h in sorted(per_stage.items(), key=lambda kv: -sum(kv[1].values())):
            valu = sum(v for g, v in h.items() if g in VALU_GROUPS)
            f64 = sum(v for g, v in h.items() if g in FP64_ARITH)
            w.writerow([st, sum(h.values()), valu, f64, f"{f64 / valu:.3f}" if valu else "", *[h.get(g, 0) for g in allg]])
    # ---- innermost loops
    trips = json.load(open(a.trips)) if a.trips else {}
    rows = []
    for k, (lo, hi) in enumerate(inner):
        h = histogram(insts, stages, lo, hi)
        st = collections.Counter(stages[lo:hi + 1]).most_common(3)
        valu = sum(v for g, v in h.items() if g in VALU_GROUPS)
        f64 = sum(v for g, v in h.items() if g in FP64_ARITH)
        rows.append(dict(loop=k, first_block=insts[lo][0], insts=hi - lo + 1, valu=valu, fp64_arith=f64,
                         share=(f64 / valu if valu else 0.0), stage="|".join(f"{n}:{c}" for n, c in st), hist=h,
                         trips=trips.get(str(k), "")))
    with open(f"{a.out}_{a.kernel}_loops.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["loop", "first_block", "instructions_per_trip", "valu", "fp64_arith", "fp64_arith_share_of_valu", "salu", "lds",
                    "stage_mix", "trips_per_wave", *allg])
        for r in rows:
            h = r["hist"]
            w.writerow([r["loop"], r["first_block"], r["insts"], r["valu"], r["fp64_arith"], f"{r['share']:.3f}",
                        h.get("salu", 0) + h.get("branch", 0) + h.get("waitcnt", 0), h.get("lds", 0), r["stage"], r["trips"],
                        *[h.get(g, 0) for g in allg]])
    tot = collections.Counter()
    for h in per_stage.values():
        tot.update(h)
    valu = sum(v for g, v in tot.items() if g in VALU_GROUPS)
    print(f"{a.kernel} = {KERNELS[a.kernel]}: {len(insts)} instructions, {len(labels)} blocks, {len(back)} loops ({len(inner)} innermost)")
    print("static VALU mix:", {g: tot[g] for g in allg if g in VALU_GROUPS and tot[g]}, "valu", valu)
    print("innermost loops with FP64 arithmetic (per trip):")
    for r in rows:
        if r["fp64_arith"] >= 4:
            h = r["hist"]
            nz = {g: v for g, v in h.items() if v and g in VALU_GROUPS and g not in FP64_ARITH}
            print(f"  loop {r['loop']:3d} {r['first_block']:12s} {r['insts']:4d} insts, valu {r['valu']:3d}, fp64 {r['fp64_arith']:3d} ({r['share']:.2f})"
                  f" rcp {h.get('fp64_trans', 0)} salu {h.get('salu', 0) + h.get('branch', 0)} wait {h.get('waitcnt', 0)} lds {h.get('lds', 0)} "
                  f"non-arith {nz}  [{r['stage']}]")


if __name__ == "__main__":
    main()
