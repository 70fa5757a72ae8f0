#!/usr/bin/env python3
"""Benchmark of the MODM + CALCTMR + RTM hot path on MI355X (BASELINE.json metric:
(wavenumber x layer x line) optical-depth evaluations per second; profiles per second).

    python bench.py --gpus N --steps K --warmup W [--workload c4|c4shard|c2|c2lc|c4brd|c2real|c3|c5|c5full] [--real-kind 8|4]

One process per GPU.  With N > 1 and no WORLD_SIZE in the environment this process only LAUNCHES the N ranks
(`python -m torch.distributed.run --nproc-per-node N bench.py ...`, before anything here touches a GPU), relays rank
0's JSON line and returns the children's exit code.  A "step" is one pass of the hot path over the rank's resident
batch of profiles: lines kernel, continuum/cloud/total kernel, rtm kernel (+ for N > 1 the single RCCL gather of the
spectral outputs to rank 0).  Inputs are resident in HBM before the timed region starts.  Rank 0 prints ONE JSON line.

Workloads (synthetic, seeded):
  c4       default headline = BASELINE configs[3] LITERALLY: 1024 sonde-like profiles x 64 layers x 50 channels x 500 lines,
           f64, split into contiguous blocks of 1024 / N profiles over the N GPUs (strong scaling: N = 1 runs all 1024)
  c4shard  the 128-profile share one GPU holds when configs[3] runs on 8 ("value_weak_shard" of the N = 1 line)
  c2       configs[1] literally: ONE profile per step (launch-latency bound)
  c2lc     the c4shard batch with every O2 line first-order line-coupled (the 60 GHz complex monoRTM exists for) and a
           model top at 0.004 hPa, where Doppler widths matter and the Voigt / speed-dependent Voigt shapes are live
  c4brd    the c4shard batch with IBRD = 1: species-by-species broadening data (the kernel instantiation with 3 waves / SIMD)
  c2real   32 profiles x 64 layers x 40 sounder channels on a line file shaped like a real aer_v_3.x product (5300 records in 23
           blocks, clustered O3 forest, O2 60-GHz complex with coupling pairs, isotopologues 1-5): one number NOT on uniform-random lines
  c3       configs[2]: 1 profile x 64 layers x 10000-wavenumber grid x 100000 lines (largest single-GPU config)
  c5full   configs[4] whole: 256 profiles with a liquid-water cloud layer, EACH viewed upwelling (IRT 1, boundary 290 K, emissivity
           0.6) and downwelling (IRT 3) = 512 runs x 64 layers x 200 channels U(0.3, 6.5) cm-1 (<= 195 GHz, the range of the TKC
           cloud model) x 500 lines, single precision (real_kind 4: the reference's "sgl" build)
  c5       its 8-GPU share: 32 profiles x 2 views
At N = 1 the line carries the headline plus, under "workloads", c3 / c5 / c2lc / c4brd / the single profile, each timed for
>= --min-seconds with its own kernel split and counter-derived roofline.

Roofline: the line sum is bound by the FP64 vector ALU, not by HBM.  "roofline" prices the FP64
work the kernel actually issued - SQ_INSTS_VALU_{FMA,ADD,MUL,TRANS}_F64 from rocprofv3 --pmc, collected LIVE by
re-running this script's workloads as a profiled child process - against the 78.6 TFLOP/s vector peak, and reports the
measured HBM bytes per launch as "traffic".  The north-star streaming model (44 B per eval against 8 TB/s) is kept as
"roofline_hbm_model"; its fraction exceeds 1 because every line record is staged once per (layer, tile) and reused
from LDS, so it is not a bound on this kernel.
"""
from __future__ import annotations

import argparse
import csv
import glob
import hashlib
import json
import os
import re
import shutil
import socket
import subprocess
import sys
import tempfile
import time
from collections import defaultdict

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BYTES_PER_EVAL = 44.0     # VNU f64 + 9 x 4-byte fields of a TAPE3 line record
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
FP64_PEAK_TFLOPS = 78.6   # vector FP64: 256 CUs x 4 SIMDs x 16 FMA lanes x 2 flop x 2.4 GHz
FP64_SUSTAINED_TFLOPS = 51.6  # measured: a pure v_fma_f64 stream, four waves per SIMD (tools/valu_cost.hip: 2.54 ns per wave instruction)
FP32_PEAK_TFLOPS = 157.3
R5_FLOP_PER_EVAL = 22.86   # FP64 flops per counted eval issued by round 5's lines_kernel<double,1,1> on configs[3] (PMC)
N_SIMD = 1024

# rocprofv3 --pmc passes (SQ: 8 slots per pass; FETCH_SIZE / WRITE_SIZE cannot share a pass; MI355X_MICROARCH.md)
PMC_PASSES = [
    ["SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_TRANS_F64", "SQ_INSTS_VALU",
     "SQ_ACTIVE_INST_VALU", "SQ_INSTS_SALU", "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE", "FETCH_SIZE"],
    ["SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_TRANS_F32", "SQ_WAIT_INST_ANY",
     "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "WRITE_SIZE"],
    # live lanes: thread-cycles of the VALU (cycles x lanes whose EXEC bit is set) against its busy cycles x 64
    ["SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64", "SQ_INSTS_VALU_CVT",
     "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"],
]
# counter families: a family sums every kernel of a step whose name contains one of its patterns.  The HIP events of
# "lines" bracket physics_kernel + far_plan_kernel + far_kernel (dense grids) + the line-sum kernel, those of "finish" the slice
# reduction + the finish kernel
FAMILIES = {"lines_kernel": ("lines_kernel", "lines_ms_kernel", "physics_kernel", "far_kernel", "far_plan_kernel"),
            "finish_kernel": ("finish_kernel", "finish_mw_kernel", "reduce_slices_kernel"),
            "rtm_kernel": ("rtm_kernel",)}
KERNELS = tuple(FAMILIES)
PMC_MARKER = "bessel_j0"   # a torch kernel nothing else here launches: the child puts one between the workloads


# ------------------------------------------------------------------------------------------------------------------
# workloads
# ------------------------------------------------------------------------------------------------------------------
C4_PROFILES = 1024   # BASELINE configs[3]
C5_PROFILES = 256    # BASELINE configs[4]


def block_of(total: int, world: int, rank: int):
    """contiguous block of ceil(total / world) profiles of rank `rank`"""
    per = (total + world - 1) // world
    lo = min(total, rank * per)
    return lo, min(total, lo + per)


def c5_views(ids, wn):
    """configs[4]: every profile is run twice - upwelling (IRT 1: ANGLE 180, TBOUND 290, emissivity 0.6, reflectivity 0.4, as
    example case 2) and downwelling (IRT 3) - over the same cloudy atmosphere (E = 2 x 256 x ...)"""
    from monortm_amd import synth

    return [synth.perturbed_profile(i, wn, nlay=64, cloud=True, irt=irt) for i in ids for irt in (1, 3)]


def build_workload(name: str, rank: int, per_gpu: int, world: int = 1):
    from monortm_amd import synth

    real_kind = 8
    t3kw = {}
    if name == "c4":
        # configs[3] as BASELINE states it: the 1024-profile batch, profile-sharded over the GPUs of the job
        rec = synth.synthetic_lines(500)
        wn = synth.c2_channels(50)
        lo, hi = block_of(C4_PROFILES, world, rank)
        profs = [synth.perturbed_profile(i, wn, nlay=64) for i in range(lo, hi)]
        desc = (f"configs[3] whole: {C4_PROFILES} synthetic sonde profiles x 64 layers x 50 channels x 500 lines, f64, "
                f"profile-sharded over {world} GPU(s): {hi - lo} profiles on this rank")
    elif name == "c4shard":
        rec = synth.synthetic_lines(500)
        nchan = int(os.environ.get("MONORTM_BENCH_CHANNELS", "50"))   # (tools/rounds_sweep.sh: other channel counts; never a bench line)
        wn = synth.c2_channels(nchan)
        profs = [synth.perturbed_profile(rank * per_gpu + i, wn, nlay=64) for i in range(per_gpu)]
        desc = (f"configs[1] profile (64 layers x {nchan} channels x 500 lines, f64) batched per configs[3]: "
                f"{per_gpu} profiles per GPU (the share of one of 8 GPUs)")
    elif name == "c4brd":
        # the IBRD = 1 instantiation of the line kernel (species-by-species broadening, the option the reference's release
        # notes single out as slow): c4shard with broadening data on 30 % of the (line, species) pairs
        rec = synth.synthetic_lines(500)
        rng = np.random.default_rng(11)
        n = len(rec.vnu)
        rec.brd_flg = (rng.random((n, 7)) < 0.3).astype(np.int32)
        dat = np.zeros((n, 21), np.float32)
        dat[:, 0::3], dat[:, 1::3], dat[:, 2::3] = (rng.uniform(0.03, 0.15, (n, 7)), rng.uniform(0.4, 0.8, (n, 7)),
                                                     rng.uniform(-0.004, 0.004, (n, 7)))
        rec.brd_dat = dat
        wn = synth.c2_channels(50)
        profs = [synth.perturbed_profile(rank * per_gpu + i, wn, nlay=64) for i in range(per_gpu)]
        for q in profs:
            q.ibrd = 1
        desc = (f"configs[1] shape with IBRD = 1 (species-by-species broadening data on 30 % of the line / species pairs): "
                f"{per_gpu} profiles x 64 layers x 50 channels x 500 lines, f64")
    elif name == "c5full":
        # configs[4] as SURVEY.md 8(d) states it: 200 channels U(0.3, 6.5) cm-1, both views of every profile
        rec = synth.synthetic_lines(500)
        wn = synth.c2_channels(200, lo=0.3, hi=6.5)
        lo, hi = block_of(C5_PROFILES, world, rank)
        profs = c5_views(range(lo, hi), wn)
        desc = (f"configs[4] whole: {C5_PROFILES} cloudy profiles x 2 views (upwelling + downwelling) x 64 layers x 200 channels "
                f"U(0.3, 6.5) cm-1 x 500 lines, single precision, profile-sharded over {world} GPU(s): {hi - lo} profiles = "
                f"{len(profs)} runs on this rank")
        real_kind = 4
    elif name == "c5full_r3":
        # configs[4] as round 3 timed it (channels U(0.3, 30) cm-1, the two views alternating over the profiles): kept for the
        # round-on-round comparison of the single-precision kernels only - c5full above is the workload SURVEY 8(d) defines
        rec = synth.synthetic_lines(500)
        wn = synth.c2_channels(200)
        profs = [synth.perturbed_profile(i, wn, nlay=64, cloud=True, irt=(1 if i % 2 == 0 else 3)) for i in range(C5_PROFILES)]
        desc = ("round 3's configs[4] workload: 256 cloudy profiles, views alternating, x 64 layers x 200 channels U(0.3, 30) cm-1 "
                "x 500 lines, single precision")
        real_kind = 4
    elif name == "c2":
        rec = synth.synthetic_lines(500)
        profs = [synth.c2_profile()]
        desc = "configs[1]: 1 profile x 64 layers x 50 channels x 500 lines, f64"
    elif name == "c2lc":
        rec = synth.synthetic_lines(500, lc_frac=1.0, sdep_frac=0.1)
        # 38 random channels + 12 channels within a few Doppler widths of line centres (spread over the O2, H2O and O3
        # lines below 30 cm-1): without them no channel ever comes within 100 Doppler widths (~1e-4 cm-1) of a centre and
        # the Voigt branch of modm.f90:427 would stay dead however thin the top layers are
        phys = (rec.iflg >= 0) & (rec.vnu > 0.3) & (rec.vnu < 30.0)
        cent = rec.vnu[phys][:: max(1, int(phys.sum()) // 12)][:12]
        offs = np.array([0.0, 3e-5, -6e-5, 1e-4, -2e-5, 5e-5, -1e-4, 8e-6, 0.0, -4e-5, 7e-5, 2e-5])[: len(cent)]
        wn = np.sort(np.concatenate([synth.c2_channels(50 - len(cent)), cent + offs]))
        profs = [synth.perturbed_profile(rank * per_gpu + i, wn, nlay=64, ztop_km=93.0) for i in range(per_gpu)]
        desc = (f"configs[1] shape with the O2 lines first-order coupled (IFLG 1 / -1 records), 10 % of the lines speed dependent, "
                f"the model top at 0.004 hPa and 12 of the 50 channels on line centres (Voigt / SD-Voigt live): {per_gpu} profiles x "
                f"64 layers x 50 channels x 500 lines, f64")
    elif name == "c2real":
        # the c4shard batch on a line file shaped like a real aer_v_3.x product instead of 500 uniform-random lines: clustered O3
        # forest, O2 60-GHz complex with coupling pairs, isotopologues 1-5, 23 blocks with a second header record (the file of the
        # reference-made fixture tests/golden/real_like.npz) seen through 40 sounder channels
        rec, t3kw, _ = synth.real_like_file()
        wn = synth.sounder_channels()
        per = max(1, per_gpu // 4)
        profs = [synth.perturbed_profile(rank * per + i, wn, nlay=64, ztop_km=60.0) for i in range(per)]
        desc = (f"real-file-like line list ({len(rec)} records, {rec.n_physical} lines in 23 TAPE3 blocks: O3 forest in clusters, O2 60-GHz complex "
                f"with coupling pairs, H2O / N2O / CO ladders, isotopologues 1-5, molecules 9-12 beyond NMOL) x {per} profiles x 64 layers "
                f"to 60 km x 40 sounder channels (22-229 GHz), f64")
    elif name == "c3":
        rec = synth.synthetic_lines(100000, seed=20261004)
        a = synth.standard_atmosphere(64)
        wn = 0.5 + 0.005 * np.arange(10000)
        profs = [synth.Profile(wn=wn, p=a["p"], t=a["t"], tz=a["tz"], wkl=a["wkl"], wbrodl=a["wbrodl"], clw=a["clw"],
                               irt=3, dvset=0.005)]
        desc = "configs[2]: 1 profile x 64 layers x 10000-wavenumber grid (0.5-50.495 cm-1) x 100000 lines, f64"
    elif name == "c5":
        rec = synth.synthetic_lines(500)
        wn = synth.c2_channels(200, lo=0.3, hi=6.5)
        per = max(1, per_gpu // 4)  # configs[4]: 256 profiles over 8 GPUs where configs[3] has 1024
        profs = c5_views(range(rank * per, rank * per + per), wn)
        desc = (f"configs[4], the share of one of 8 GPUs: {per} cloudy profiles x 2 views (upwelling + downwelling) x 64 layers x "
                f"200 channels U(0.3, 6.5) cm-1 x 500 lines, single precision")
        real_kind = 4
    else:
        raise SystemExit(f"unknown workload {name}")
    return rec, profs, desc, real_kind, t3kw


def evals_per_step(rt, profs) -> float:
    """E = sum_profiles NWN * sum_layers NL_layer, NL = physical line records of molecules with a
    non-zero column in the layer."""
    counts = np.array([rt.line_count(m) for m in range(1, profs[0].nmol + 1)], np.float64)
    e = 0.0
    for p in profs:
        e += p.nwn * float(((p.wkl != 0.0) * counts[None, :]).sum())
    return e


class Resident:
    """One workload resident on this rank's GPU."""

    def __init__(self, name, rank, local, per_gpu, real_kind=0, tmp=None, world=1):
        from monortm_amd import api, tape3

        self.name = name
        self.rec, self.profs, self.desc, rk, t3kw = build_workload(name, rank, per_gpu, world)
        self.real_kind = real_kind or rk
        self.tmp = tmp or tempfile.mkdtemp(prefix=f"monortm_bench_r{rank}_")
        t3 = os.path.join(self.tmp, f"TAPE3_{name}")
        tape3.write_tape3(t3, self.rec, **t3kw)
        self.rt = api.MonoRTM(t3, self.profs[0].wn[0], self.profs[0].wn[-1], device=local, real_kind=self.real_kind)
        self.batch = api.DeviceBatch(self.rt, self.profs, device=f"cuda:{local}")
        self.e_step = evals_per_step(self.rt, self.profs)

    def config(self, world=1, graph=False):
        p0 = self.profs[0]
        return {"workload": f"{self.name}: {self.desc}", "profiles_per_gpu": len(self.profs), "layers": p0.nlay,
                "wavenumbers": p0.nwn, "lines": int(self.rt.line_count(0)), "nmol": p0.nmol,
                "evals_per_step_per_gpu": self.e_step, "launch": "hip graph replay" if graph else "3 stream launches",
                "parallelism": f"profile-sharded x{world}" + (", one RCCL gather/step" if world > 1 else "")}

    def close(self):
        self.rt.close()


def warmup_rounds(step, sync, torch, dist, world, warmup, warm_seconds, device):
    """At least `warmup` untimed steps, continued until `warm_seconds` have passed (the shader clock settles during the first
    second) - with the SAME number of steps on every rank: a step of a sharded job ends in a collective (GatherPlan), so a
    rank that warmed up one step longer than its peers would leave a gather without partners and the job would hang at the
    first barrier (round 5: the time-based loop used to run per rank).  The ranks advance in rounds of 8 steps and agree after
    each round (MAX over ranks of "not warm yet") whether to go on."""
    t_w = time.perf_counter()
    n_w = 0
    while True:
        for _ in range(8 if (warm_seconds > 0 or warmup > 8) else max(warmup, 1)):
            step()
            n_w += 1
        sync()
        more = n_w < warmup or (time.perf_counter() - t_w < warm_seconds and n_w < 100000)
        if world > 1:
            flag = torch.tensor([1.0 if more else 0.0], dtype=torch.float32, device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            more = bool(flag.item() > 0)
        if not more:
            return n_w


def timed_steps(torch, dist, res: Resident, steps, warmup, warm_seconds, plan=None, graph=False, events=True, world=1, gather_every=1):
    """W warm-up steps (continued until warm_seconds have passed, so that the clock has settled), then exactly `steps`
    steps between barrier + synchronize; HIP events around the dominant kernel inside the timed region."""
    batch, rt = res.batch, res.rt
    if graph:
        batch.capture()

    nstep, gstart = [0], [0]

    def step():
        # the block a gather reads must not be overwritten while it is in flight (GatherPlan, field_major): consecutive steps
        # write alternate blocks, so the step right after a gather's start is safe and the one after that is not (only met with
        # --gather-every K > 1: with K = 1 the next start() has waited by then); a replayed graph rewrites the ONE block it was
        # captured with, so it waits every time
        if plan is not None and plan.work is not None and (graph or nstep[0] - gstart[0] >= 1):
            plan.wait()
        if graph:
            batch.replay()
        else:
            batch.step()
        nstep[0] += 1
        if plan is not None and nstep[0] % gather_every == 0:
            plan.start(batch.spectral_block())
            gstart[0] = nstep[0]

    n_w = warmup_rounds(step, torch.cuda.synchronize, torch, dist, world, warmup, warm_seconds, batch.TB.device)
    if plan is not None:
        plan.wait()
    batch.check()
    torch.cuda.synchronize()
    # HIP events around the dominant kernel inside the timed region: every stride-th launch (8 samples of 200 steps, 4 of
    # 20; the kernel's duration varies by ~1 % from launch to launch) - an event pair in the stream keeps the next launch from
    # being queued behind the running kernel (~9 us per bracketed step: 0.2301 ms per step with every 8th launch bracketed
    # against 0.2281 ms with none), and the probe should not slow down what it measures
    rt.profile(0 if (not events or graph) else 1, stride=max(1, min(25, steps // 4)))
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    if plan is not None:
        plan.wait()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rt.profile(0)
    batch.check()
    ms_lines, n_lines = rt.kernel_time(0)
    # the two small kernels (and the line kernel under graph replay) are timed in a few extra untimed steps
    # (three batches of five steps, the quietest batch counts: a single stall of the queue would otherwise sit in a mean of 5)
    best = None
    for _ in range(3):
        before = [rt.kernel_time(k) for k in range(3)]
        rt.profile(7 if (graph or not events) else 6)
        for _ in range(5):
            batch.step()
        torch.cuda.synchronize()
        rt.profile(0)
        after = [rt.kernel_time(k) for k in range(3)]
        means = [((a[0] - b[0]) / max(a[1] - b[1], 1), a[1] - b[1]) for a, b in zip(after, before)]
        if best is None or sum(m[0] for m in means) < sum(m[0] for m in best):
            best = means
    (ms_fin, n_fin), (ms_rtm, n_rtm) = best[1], best[2]
    ms_fin, ms_rtm = ms_fin * n_fin, ms_rtm * n_rtm
    if graph or not events:
        ms_lines, n_lines = best[0][0] * best[0][1], best[0][1]
    return {"dt": dt, "steps": steps, "warmup_steps_run": n_w,
            "kernel_ms": {"lines": ms_lines / max(n_lines, 1), "continuum_cloud_total": ms_fin / max(n_fin, 1),
                          "rtm": ms_rtm / max(n_rtm, 1)}, "lines_launches": n_lines}


def measure_by_duration(torch, res: Resident, min_seconds: float, graph=False):
    """Secondary workloads: probe the step time, then time exactly n = ceil(min_seconds / step) steps."""
    b = res.batch
    if graph:
        b.capture()
    run = b.replay if graph else b.step
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    est = max((time.perf_counter() - t0) / 3, 1e-6)
    n = int(min(200000, max(5, np.ceil(min_seconds / est))))
    m = timed_steps(torch, None, res, n, 3, 0.3, graph=graph)
    return m


# ------------------------------------------------------------------------------------------------------------------
# counters: rocprofv3 --pmc on a child run of the same workloads
# ------------------------------------------------------------------------------------------------------------------
def csrc_hash() -> str:
    h = hashlib.sha1()
    base = os.path.join(ROOT, "monortm_amd", "csrc")
    for dp, _, files in sorted(os.walk(base)):
        for f in sorted(files):
            if f.endswith((".hip", ".hpp", ".cpp", ".h")):
                h.update(f.encode())
                h.update(open(os.path.join(dp, f), "rb").read())
    return h.hexdigest()[:16]


def pmc_child(args):
    """Run under rocprofv3 --pmc: a few steps of every requested workload, in order; the manifest tells the parent how
    many launches of each kernel belong to which workload."""
    import torch

    torch.cuda.set_device(0)
    manifest = []
    tmp = tempfile.mkdtemp(prefix="monortm_pmc_")
    mark = torch.ones(64, device="cuda")
    for name in args.pmc_workloads.split(","):
        res = Resident(name, 0, 0, args.profiles_per_gpu, tmp=tmp)
        nsteps = 3
        torch.cuda.synchronize()
        torch.special.bessel_j0(mark)   # marker dispatch: the steps of this workload follow it
        for _ in range(nsteps):
            res.batch.step()
        torch.cuda.synchronize()
        torch.special.bessel_j0(mark)   # ... and end here
        res.batch.check()
        manifest.append({"workload": name, "steps": nsteps})
        res.close()
    with open(args.pmc_manifest, "w") as f:
        json.dump(manifest, f)
    shutil.rmtree(tmp, ignore_errors=True)


def collect_pmc(workloads, per_gpu, timeout_s=240):
    """-> {workload: {kernel: {counter: mean per launch}}} measured now, or raises."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        raise RuntimeError("rocprofv3 not found")
    env = {k: v for k, v in os.environ.items()
           if not (k.startswith(("ROCPROF", "ROCP_", "ROCTX", "HSA_TOOLS")) or k in ("LD_PRELOAD", "RANK", "WORLD_SIZE", "LOCAL_RANK"))}
    env["TMPDIR"] = "/tmp"
    out = {w: {k: {} for k in KERNELS} for w in workloads}
    work = tempfile.mkdtemp(prefix="monortm_pmcrun_", dir="/tmp")
    try:
        for ip, counters in enumerate(PMC_PASSES):
            d = os.path.join(work, f"pass{ip}")
            man = os.path.join(work, f"manifest{ip}.json")
            cmd = [exe, "--pmc", *counters, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__),
                   "--pmc-child", "--pmc-workloads", ",".join(workloads), "--pmc-manifest", man, "--profiles-per-gpu", str(per_gpu)]
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
            if r.returncode != 0 or not os.path.exists(man):
                if ip >= 2:  # the optional passes (live lanes, instruction classes) may name counters a rocprofv3 build lacks
                    out["_failed_passes"] = out.get("_failed_passes", []) + [ip]
                    continue
                raise RuntimeError(f"rocprofv3 pass {ip} failed (rc {r.returncode}): {(r.stderr or r.stdout)[-300:]}")
            manifest = json.load(open(man))
            # dispatch id -> (kernel name, {counter: value summed over the rows of that dispatch})
            disp = {}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        e = disp.setdefault(int(row["Dispatch_Id"]), [row["Kernel_Name"], defaultdict(float)])
                        e[1][row["Counter_Name"]] += float(row["Counter_Value"])
            ids = sorted(disp)
            marks = [i for i in ids if PMC_MARKER in disp[i][0]]
            if len(marks) != 2 * len(manifest):
                raise RuntimeError(f"pass {ip}: {len(marks)} marker dispatches for {len(manifest)} workloads")
            for k, m in enumerate(manifest):  # the dispatches between the k-th pair of markers are this workload's steps
                lo, hi = marks[2 * k], marks[2 * k + 1]
                for fam, pats in FAMILIES.items():
                    tot, n = defaultdict(float), 0
                    for i in ids:
                        if lo < i < hi and any(pt in disp[i][0] for pt in pats):
                            n += 1
                            for cname, v in disp[i][1].items():
                                tot[cname] += v
                    if n % m["steps"] != 0:
                        raise RuntimeError(f"{m['workload']}/{fam}: {n} dispatches in {m['steps']} steps")
                    for cname, v in tot.items():
                        out[m["workload"]][fam][cname] = v / m["steps"]     # per step (= per launch of the family)
                    out[m["workload"]][fam]["_dispatches_per_step"] = n / m["steps"]
                    # which kernels of the family ran (lines_kernel or lines_ms_kernel: the lane layouts differ)
                    out[m["workload"]][fam]["_kernels"] = sorted({(re.search(r"(\w+_kernel)\b", disp[i][0]) or [disp[i][0]])[0] for i in ids
                                                                 if lo < i < hi and any(pt in disp[i][0] for pt in pats)})
    finally:
        shutil.rmtree(work, ignore_errors=True)
    return out


def channel_lane_frac(nwn: int, ms: bool = False) -> float:
    """Share of the wavenumber lanes of lines_kernel's tiles that hold a wavenumber (tile widths of lines_config(), api.hip): 50
    channels occupy 50 of the 64 lanes of a one-wave tile.  The idle lanes run with EXEC set (their results are discarded), so
    the EXEC-based live_lane_frac does not see them.  ms: lines_ms_kernel - G states x ceil(nwn / 5) lanes, five wavenumbers a
    lane (50 channels: 6 states x 10 lanes x 5 = 300 of the 320 evaluation slots of a wave)."""
    if ms and nwn <= 64:
        lps = (nwn + 4) // 5
        return min(12, 64 // lps) * nwn / (64.0 * 5.0)
    tile = 64 if nwn <= 64 else 128 if nwn <= 256 else 512
    return nwn / (tile * ((nwn + tile - 1) // tile))


_PK_SHARE = None


def pk_f32_share() -> float:
    """Share of the FP32 FMA / MUL / ADD instructions of the single-precision evaluate loops that are packed (v_pk_*_f32), from the
    newest committed ISA census of lines_kernel<float,1,4> (profiles/r*_isa_census_f14_loops.csv, tools/isa_census.py); 0.9 if none."""
    global _PK_SHARE
    if _PK_SHARE is None:
        _PK_SHARE = 0.9
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_isa_census_f14_loops.csv")), reverse=True):
            try:
                pk = sc = 0
                for row in csv.DictReader(open(f)):
                    if "eval_" in row.get("stage_mix", ""):
                        pk += int(row.get("fp32_pk") or 0)
                        sc += int(row.get("fp32_arith") or 0)
                if pk + sc > 0:
                    _PK_SHARE = pk / (pk + sc)
                    break
            except Exception:
                pass
    return _PK_SHARE


def roofline_from_counters(c: dict | None, avg_ms: float, e_step: float, source: str | None, f32: bool = False, nwn: int = 0):
    """The bound that holds for the line sum: FP64 vector ALU.  flop = 64 lanes x (2 FMA + ADD + MUL + TRANS) wave
    instructions (masked lanes included - see fp64_pipe_util for the slot view)."""
    r = {"bound": "valu_fp64", "kernel": "lines_kernel", "achieved": None, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
         "frac": None, "traffic": None, "avg_launch_ms": avg_ms, "counter_source": source}
    if not c or avg_ms <= 0:
        return r
    g = lambda k: float(c.get(k, 0.0))  # noqa: E731
    secs = avg_ms * 1e-3
    f64 = {k: g(f"SQ_INSTS_VALU_{k}_F64") for k in ("FMA", "ADD", "MUL", "TRANS")}
    flop64 = 64.0 * (2 * f64["FMA"] + f64["ADD"] + f64["MUL"] + f64["TRANS"])
    r["achieved"] = flop64 / secs / 1e12
    r["frac"] = r["achieved"] / FP64_PEAK_TFLOPS
    r["fp64_flop_per_launch"] = flop64
    r["fp64_flop_per_eval"] = flop64 / e_step if e_step else None
    if e_step and not f32 and nwn and nwn <= 64:
        # frac prices the FP64 work the kernel ISSUED; a kernel that removes work (round 6: four lines per reciprocal, slots of
        # channels out of a line's reach skipped, one prologue for six states) lowers it while it gets faster.  The same evals at
        # the arithmetic lines_kernel<double,1,1> issues per eval on configs[3] (22.86 flop, profiles/r05_z_pmc_per_launch.json):
        r["r5_flop_per_eval"] = R5_FLOP_PER_EVAL
        r["frac_at_r5_flop_per_eval"] = R5_FLOP_PER_EVAL * e_step / secs / 1e12 / FP64_PEAK_TFLOPS
    f32c = {k: g(f"SQ_INSTS_VALU_{k}_F32") for k in ("FMA", "ADD", "MUL", "TRANS")}
    # SQ_INSTS_VALU_{FMA,ADD,MUL}_F32 count a PACKED instruction (v_pk_fma_f32: two FP32 operations per lane) ONCE - measured:
    # tools/pk_count.hip, profiles/r05_pk_count_counters.csv (4096 v_pk_fma_f32 per wave read 4096, like 4096 v_fma_f32).  The
    # single-precision line sum issues nearly all of its FP32 arithmetic packed (pk_share: from the ISA census of the timed
    # instantiation), so the operations are (1 + pk_share) x the counted instructions; v_rcp_f32 (TRANS) is never packed.
    pk = pk_f32_share() if f32 else 0.0
    flop32 = 64.0 * ((2 * f32c["FMA"] + f32c["ADD"] + f32c["MUL"]) * (1.0 + pk) + f32c["TRANS"])
    if flop32:
        r["fp32_tflops"] = flop32 / secs / 1e12
        if f32:
            r["fp32_packed_share_of_counted_insts"] = pk
            r["fp32_counting"] = ("SQ_INSTS_VALU_*_F32 count a packed instruction once (tools/pk_count.hip); flops = counted x (1 + packed share "
                                  "of the evaluate loops, tools/isa_census.py)")
    if f32:  # single-precision build: float Lorentz loops + double prepare stage; both pipes priced
        r["bound"] = "valu_fp32+fp64"
        r["frac"] = r["achieved"] / FP64_PEAK_TFLOPS + flop32 / secs / 1e12 / FP32_PEAK_TFLOPS
    # what the vector ALU sustains: a pure v_fma_f64 stream delivers 51.6 of the nominal 78.6 TFLOP/s on this chip (the shader clock
    # drops under FP64 load; tools/valu_cost.hip, LABNOTES section 5) - the headroom of the kernel is against THAT
    r["sustained_peak"] = FP64_SUSTAINED_TFLOPS
    r["frac_of_sustained"] = r["achieved"] / FP64_SUSTAINED_TFLOPS + (flop32 / secs / 1e12 / FP32_PEAK_TFLOPS if f32 else 0.0)
    cyc = g("GRBM_GUI_ACTIVE") / 8.0  # the counter sums the 8 XCDs
    if cyc > 0:
        r["shader_clock_ghz"] = cyc / secs / 1e9
        n64 = sum(f64.values())
        # a wave64 FP64 instruction occupies its SIMD's FP64 pipe for 4 cycles (16 lanes per clock)
        r["fp64_pipe_util"] = 4.0 * n64 / (N_SIMD * cyc)
        r["valu_busy"] = 4.0 * g("SQ_ACTIVE_INST_VALU") / (N_SIMD * cyc) if g("SQ_ACTIVE_INST_VALU") else None
    if g("SQ_THREAD_CYCLES_VALU") and g("SQ_ACTIVE_INST_VALU"):
        # frac counts 64 lanes per FP64 instruction whatever EXEC holds; the share of VALU lane-cycles with the EXEC bit set
        # (all VALU instructions: the counters do not split it by type) scales it to the lanes that did work
        # (rocprofv3's own derived metric VALUUtilization: THREAD_CYCLES_VALU / (ACTIVE_INST_VALU x wave size))
        r["live_lane_frac"] = g("SQ_THREAD_CYCLES_VALU") / (64.0 * g("SQ_ACTIVE_INST_VALU"))
        r["frac_live_lanes"] = r["frac"] * r["live_lane_frac"]
    if nwn:
        # ... and the lanes of a wavenumber tile that hold no wavenumber (applied to the whole kernel: a lower bound, the
        # prepare stage - one lane per line - is not affected)
        kern = c.get("_kernels") or []
        r["kernels"] = kern
        if kern:  # the name rocprofv3 lists: lines_ms_kernel for a batch of a round of waves or more, lines_kernel otherwise
            r["kernel"] = "+".join(sorted(set(kern)))
        r["channel_lane_frac"] = channel_lane_frac(nwn, ms=any("lines_ms_kernel" in k for k in kern))
        r["frac_useful_lanes"] = r["frac"] * r.get("live_lane_frac", 1.0) * r["channel_lane_frac"]
    if g("SQ_INSTS_VALU"):
        for k in ("INT32", "INT64", "CVT"):
            if g(f"SQ_INSTS_VALU_{k}"):
                r[f"{k.lower()}_share_of_valu_insts"] = g(f"SQ_INSTS_VALU_{k}") / g("SQ_INSTS_VALU")
        if f32 or flop32:
            r["f32_share_of_valu_insts"] = sum(f32c.values()) / g("SQ_INSTS_VALU")
        r["dispatches_per_step"] = c.get("_dispatches_per_step")
        r["valu_insts_per_launch"] = g("SQ_INSTS_VALU")
        r["f64_share_of_valu_insts"] = sum(f64.values()) / g("SQ_INSTS_VALU")
        r["salu_per_valu"] = g("SQ_INSTS_SALU") / g("SQ_INSTS_VALU")
    if g("SQ_LDS_IDX_ACTIVE"):
        r["lds_bank_conflict_frac"] = g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        # rocprofv3 reports KB; gfx950 counts a wide read at half its bytes (MI355X_MICROARCH.md, HBM): doubled = upper bound
        r["traffic"] = (2.0 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024.0
        r["hbm_measured_gbs"] = r["traffic"] / secs / 1e9
        r["hbm_measured_frac"] = r["hbm_measured_gbs"] / HBM_PEAK_GBS
    return r


def hbm_model(avg_ms, e_step):
    ach = BYTES_PER_EVAL * e_step / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    return {"bound": "hbm (north-star streaming model, NOT a bound on this kernel)", "achieved": ach, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": BYTES_PER_EVAL * e_step,
            "note": "44 B per counted eval as if every (wavenumber, layer, line) visit re-read its TAPE3 record; the kernel stages a "
                    "record once per (layer, tile) and reuses it from LDS for every wavenumber, and never visits lines outside "
                    "the 25 cm-1 window, so this fraction exceeds 1 and prices nothing - see roofline (FP64 VALU) and its "
                    "measured hbm traffic"}


# ------------------------------------------------------------------------------------------------------------------
# CPU baseline: the reference itself on the host cores
# ------------------------------------------------------------------------------------------------------------------
def _big_stack():
    """The reference keeps its per-wavenumber work arrays on the stack (a 10000-wavenumber call overflows the default 8 MB)."""
    import resource

    try:
        resource.setrlimit(resource.RLIMIT_STACK, (resource.RLIM_INFINITY, resource.RLIM_INFINITY))
    except Exception:
        pass


def cpu_baseline(rec, profs, nsample: int, sgl: bool = False, census_profile=None, ncpu_procs: int = 16):
    """Time the reference itself (oracle/_ref/harness_ref_dbl_fast - or harness_ref_sgl_fast, its "sgl" flag set, for the
    single-precision workload: the reference's own sources compiled by amdflang, hot-path units at -O2) on a bounded sample
    of the same workload, 1 host core (the reference is serial).  Falls back to the C restatement (kind "port") if the
    prebuilt binary is absent."""
    from monortm_amd import caseio, tape3

    sample = profs[:nsample]
    census = None
    counts = {}
    r = np.asarray(rec.mol[rec.iflg >= 0]) % 100
    for m in np.unique(r):
        counts[int(m)] = int((r == m).sum())
    ev = sum(p.nwn * sum(counts.get(m + 1, 0) * int((p.wkl[:, m] != 0).sum()) for m in range(p.nmol)) for p in sample)
    harness = os.path.join(ROOT, "oracle", "_ref", "harness_ref_sgl_fast" if sgl else "harness_ref_dbl_fast")
    build = 'the "sgl" build (default REAL = 4 bytes)' if sgl else 'the "dbl" build'
    with tempfile.TemporaryDirectory() as d:
        tp, cp, op = (os.path.join(d, n) for n in ("TAPE3", "case.bin", "out.bin"))
        tape3.write_tape3(tp, rec)
        try:  # cut-pass fraction and Lorentz / Voigt split, counted by the C restatement on one profile
            from oracle.pyoracle import Oracle

            cpr = census_profile if census_profile is not None else sample[0]   # (c3: a thinned grid - the census is a ratio)
            orc = Oracle(tp, cpr.wn[0], cpr.wn[-1])
            orc.census(reset=True)
            orc.run(cpr)
            c = orc.census()
            orc.close()
            census = {"profile": 0, "line_visits": c["visits"], "cut_pass_frac": 1.0 - c["cut_rejected"] / max(c["visits"], 1),
                      "lorentz_frac": c["lorentz"] / max(c["lorentz"] + c["voigt"], 1),
                      "voigt_frac": c["voigt"] / max(c["lorentz"] + c["voigt"], 1)}
        except Exception as e:  # the census is informative only
            census = {"error": str(e)}
        if os.path.exists(harness):
            caseio.write_case(cp, sample)
            t0 = time.perf_counter()
            r = subprocess.run([harness, cp, tp, op], cwd=d, capture_output=True, text=True, preexec_fn=_big_stack)
            wall = time.perf_counter() - t0
            secs = None
            for line in r.stdout.splitlines():
                if line.startswith("HARNESS_SECONDS"):
                    secs = float(line.split()[1])
            if r.returncode == 0 and secs:
                allc = None
                try:
                    # the same sample split over this GPU's share of the host cores (16 per GPU on the pool's boxes; --cpu-procs),
                    # one reference process per core (the program is serial), >= 16 profiles each.  Every process times its own
                    # MODM + CALCTMR + RTM calls (HARNESS_SECONDS, as the 1-core figure does), so program start-up and the TAPE3
                    # load are NOT in the rate: round 5 divided by the wall clock of 256 processes x ONE profile, where start-up
                    # was 70x the work.  rate = all evals / the slowest process's in-harness seconds.
                    nc = max(1, min(len(sample) // 16, len(os.sched_getaffinity(0)), ncpu_procs))
                    if nc > 1:
                        procs = []
                        t1 = time.perf_counter()
                        for k in range(nc):
                            ck, ok, lk = (os.path.join(d, f"{n}{k}") for n in ("case", "out", "log"))
                            caseio.write_case(ck, sample[k::nc])
                            procs.append((subprocess.Popen([harness, ck, tp, ok], cwd=d, stdout=open(lk, "w"),
                                                           stderr=subprocess.DEVNULL, preexec_fn=_big_stack), lk))
                        rcs = [p.wait() for p, _ in procs]
                        w_all = time.perf_counter() - t1
                        hs = []
                        for _, lk in procs:
                            for line in open(lk):
                                if line.startswith("HARNESS_SECONDS"):
                                    hs.append(float(line.split()[1]))
                        if all(rc == 0 for rc in rcs) and len(hs) == nc:
                            allc = {"value": ev / max(hs), "unit": "evals/s", "cores": nc, "profiles_per_process": len(sample) // nc,
                                    "slowest_process_s": max(hs), "wall_s": w_all,
                                    "note": f"{nc} reference processes side by side, {len(sample) // nc} profiles each; rate = all evals / the "
                                            f"slowest process's in-harness seconds (start-up and TAPE3 load excluded, as in the 1-core figure)"}
                except Exception as e:
                    allc = {"error": str(e)}
                return {"value": ev / secs, "unit": "evals/s", "cores": 1, "kind": "reference", "census": census, "all_cores": allc,
                        "ms_per_profile": secs / len(sample) * 1e3,
                        "sample": f"{len(sample)} profile(s) of the workload = {ev:.3g} evals in {secs:.2f} s "
                                  f"(MODM+CALCTMR+RTM inside the reference, wall {wall:.2f} s; {build}, amdflang, hot path -O2)"}
        from oracle.pyoracle import Oracle

        orc = Oracle(tp, sample[0].wn[0], sample[0].wn[-1])
        t0 = time.perf_counter()
        for p in sample:
            orc.run(p)
        secs = time.perf_counter() - t0
        return {"value": ev / secs, "unit": "evals/s", "cores": 1, "kind": "port", "census": census,
                "sample": f"{len(sample)} profile(s) = {ev:.3g} evals in {secs:.2f} s (oracle/monortm_oracle.c, gcc -O2)"}


def dropin_latency(rec, profs, tmp):
    """The reference's own calling pattern through the drop-in modules: one profile per MODM / CALCTMR / RTM call
    (src/monortm.f90:557-574), host arrays in and out - examples/harness.f90 linked against monortm_amd/fortran (the same
    program the parity tests link against the reference).  PCIe-inclusive; never the headline value."""
    from monortm_amd import _build, caseio, tape3

    exe = os.path.join(_build.LIBDIR, "harness_hip_dbl")
    if not os.path.exists(exe):
        return {"error": "monortm_amd/lib/harness_hip_dbl not built"}
    tp, cp, op = (os.path.join(tmp, n) for n in ("TAPE3_dropin", "case_dropin.bin", "out_dropin.bin"))
    tape3.write_tape3(tp, rec)
    caseio.write_case(cp, profs)
    best = None
    for _ in range(2):
        r = subprocess.run([exe, cp, tp, op, "3"], cwd=tmp, capture_output=True, text=True, timeout=300)
        if r.returncode != 0:
            return {"error": (r.stdout + r.stderr)[-300:]}
        for line in r.stdout.splitlines():
            if line.startswith("HARNESS_FIRST_CALL"):
                w = line.split()
                first, st = float(w[1]), [float(x) for x in w[-3:]]
                if best is None or sum(st) < sum(best[1]):
                    best = (first, st)
    if best is None:
        return {"error": "harness printed no HARNESS_FIRST_CALL line: " + r.stdout[-200:]}
    n = 3 * len(profs) - 1  # every call after the first one (which loads TAPE3 and starts the device, like GET_LNFL in the reference)
    first, st = best
    return {"ms_per_profile": sum(st) / n * 1e3, "ms_modm": st[0] / n * 1e3, "ms_calctmr": st[1] / n * 1e3, "ms_rtm": st[2] / n * 1e3,
            "first_call_ms": first * 1e3, "profiles": len(profs), "repeats": 3, "best_of": 2,
            "what": "MODM + CALCTMR + RTM through the ISO_C_BINDING drop-in modules, one profile per call, host arrays in/out "
                    "(PCIe-inclusive, steady state after the first call, the faster of two runs of the program; first_call_ms = device start-up + TAPE3 load + first "
                    "profile); examples/harness.f90 = the call sequence of src/monortm.f90:557-574"}


# ------------------------------------------------------------------------------------------------------------------
# output: the full record goes to gpurun_out/bench_detail.json and to stderr; stdout carries ONE compact line the driver can
# keep whole (round 5's single 22 KB line outgrew the driver's capture: "parsed": null)
# ------------------------------------------------------------------------------------------------------------------
COMPACT_LIMIT = 4096


def _strict(x):
    """strict JSON: non-finite floats become null, numpy scalars become Python numbers"""
    if isinstance(x, dict):
        return {str(k): _strict(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_strict(v) for v in x]
    if isinstance(x, (np.floating, np.integer)):
        x = x.item()
    if isinstance(x, float) and not np.isfinite(x):
        return None
    return x


def _sig(x, n=6):
    if isinstance(x, float):
        return float(f"{x:.{n}g}")
    if isinstance(x, dict):
        return {k: _sig(v, n) for k, v in x.items()}
    if isinstance(x, list):
        return [_sig(v, n) for v in x]
    return x


def compact_line(out: dict) -> dict:
    """The headline keys of the contract + one (value, ms_per_step, frac) triple per secondary workload; <= COMPACT_LIMIT bytes."""
    keep = ("metric", "value", "unit", "n_gpus", "n_ranks_seen", "steps", "warmup", "ms_per_step", "timed_ms", "higher_is_better",
            "scaling", "vs_baseline", "dtype", "data", "profiles_per_sec", "kernel_ms_per_step", "shapes_evaluated_per_s",
            "value_weak_shard", "gather_every", "gather_us_alone", "gathers_in_timed_region", "stub")
    c = {k: out[k] for k in keep if k in out}
    cfg = dict(out.get("config", {}))
    if len(str(cfg.get("workload", ""))) > 200:
        cfg["workload"] = cfg["workload"][:197] + "..."
    c["config"] = cfg
    r = out.get("roofline")
    if r:
        rk = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "launches", "counter_source",
              "sustained_peak", "frac_of_sustained", "frac_useful_lanes", "live_lane_frac", "channel_lane_frac", "valu_busy", "kernels", "frac_at_r5_flop_per_eval",
              "fp64_flop_per_eval", "hbm_measured_gbs", "salu_per_valu")
        c["roofline"] = {k: r[k] for k in rk if k in r}
        src = c["roofline"].get("counter_source")
        if isinstance(src, str) and len(src) > 120:
            c["roofline"]["counter_source"] = src[:117] + "..."
    h = out.get("roofline_hbm_model")
    if h:
        c["roofline_hbm_model"] = {"achieved": h.get("achieved"), "peak": h.get("peak"), "unit": h.get("unit"), "frac": h.get("frac"),
                                   "note": "44 B/eval streaming model: NOT a bound (records are reused from LDS)"}
    b = out.get("cpu_baseline")
    if b:
        c["cpu_baseline"] = {k: b[k] for k in ("value", "unit", "cores", "kind", "sample", "ms_per_profile", "error") if k in b}
        smp = c["cpu_baseline"].get("sample")
        if isinstance(smp, str) and len(smp) > 220:
            c["cpu_baseline"]["sample"] = smp[:217] + "..."
        a = b.get("all_cores")
        if isinstance(a, dict) and a.get("value"):
            c["cpu_baseline"]["all_cores"] = {k: a[k] for k in ("value", "cores", "profiles_per_process", "slowest_process_s") if k in a}
        cen = b.get("census")
        if isinstance(cen, dict) and "cut_pass_frac" in cen:
            c["cpu_baseline"]["cut_pass_frac"] = cen["cut_pass_frac"]
            c["cpu_baseline"]["voigt_frac"] = cen.get("voigt_frac")
    w = out.get("workloads")
    if w:
        c["workloads"] = {}
        for name, x in w.items():
            if not isinstance(x, dict) or "error" in x:
                c["workloads"][name] = {"error": str(x.get("error"))[:80] if isinstance(x, dict) else "?"}
                continue
            y = {"value": x.get("value"), "ms_per_step": x.get("ms_per_step"), "dtype": x.get("dtype"),
                 "lines_ms": (x.get("kernel_ms_per_step") or {}).get("lines")}
            if isinstance(x.get("roofline"), dict):
                y["frac"] = x["roofline"].get("frac")
                if x["roofline"].get("traffic") is not None:
                    y["traffic"] = x["roofline"]["traffic"]
            if isinstance(x.get("cpu_baseline"), dict) and x["cpu_baseline"].get("value"):
                y["cpu_value"] = x["cpu_baseline"]["value"]
            c["workloads"][name] = y
    d = out.get("dropin")
    if isinstance(d, dict) and "ms_per_profile" in d:
        c["dropin_ms_per_profile"] = d["ms_per_profile"]
    if "detail_file" in out:
        c["detail_file"] = out["detail_file"]
    c = _sig(_strict(c))
    # never over the limit: drop the optional parts, least important first
    for k in ("workloads", "roofline_hbm_model", "kernel_ms_per_step", "dropin_ms_per_profile"):
        if len(json.dumps(c, allow_nan=False)) <= COMPACT_LIMIT - 200:
            break
        c.pop(k, None)
    return c


def emit(out: dict, detail_file: str = ""):
    """stdout carries ONE line, the compact object (the driver's capture of stdout is bounded and its parser unknown: nothing else
    is printed there); the full record goes to gpurun_out/bench_detail.json (gpurun merges that directory back) and to stderr."""
    full = _strict(out)
    path = detail_file or os.path.join(ROOT, "gpurun_out", "bench_detail.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(full, f, indent=1, allow_nan=False)
        full["detail_file"] = out["detail_file"] = os.path.relpath(path, ROOT)
    except OSError:
        pass
    sys.stderr.write("BENCH_DETAIL " + json.dumps(full, allow_nan=False) + "\n")
    sys.stderr.flush()
    line = json.dumps(compact_line(out), allow_nan=False)
    assert len(line) < COMPACT_LIMIT, len(line)
    print(line)
    sys.stdout.flush()


# ------------------------------------------------------------------------------------------------------------------
# launcher: N ranks as fresh child processes
# ------------------------------------------------------------------------------------------------------------------
def launch_ranks(n: int, argv: list[str]) -> int:
    """`python bench.py --gpus N` without a torch.distributed.run parent: start the N ranks ourselves.  Nothing in this
    process has touched a GPU (no torch.cuda call, no HIP library loaded) - the ranks are ordinary child processes."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *argv]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.run(cmd, env=env).returncode


def stub_rank(args, world, rank):
    """MONORTM_BENCH_STUB=1: the launcher / rendezvous / gather / timing skeleton with a no-op step on CPU tensors (gloo).
    Used by tests/test_bench_launcher.py to cover `bench.py --gpus N` where there is no GPU; never a measurement."""
    import torch
    import torch.distributed as dist

    from monortm_amd import distributed as D

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    per, nwn = 4, 5
    local = torch.full((6, per, nwn), float(rank), dtype=torch.float64)   # the field-major block of the real path
    plan = D.GatherPlan(per * world, local, field_major=True) if world > 1 else None
    seen = world
    if world > 1:
        t = torch.ones(1)
        dist.all_reduce(t)
        seen = int(t.item())
        dist.barrier()
    # the warm-up of the real path, collective per step included: time-based, so its length must be agreed between the ranks
    # (rank-dependent sleep: without the agreement the ranks would leave the loop after different numbers of gathers)
    def wstep():
        if plan is not None:
            plan.start(local)
        time.sleep(0.002 * (1 + rank))

    n_w = warmup_rounds(wstep, lambda: None, torch, dist, world, args.warmup, min(args.min_seconds, 0.2), torch.device("cpu"))
    if plan is not None:
        plan.wait()
        dist.barrier()
    t0 = time.perf_counter()
    ngather = 0
    for i in range(args.steps):
        if plan is not None and (i + 1) % max(1, args.gather_every) == 0:   # (--gather-every K: the collective of every K-th step only)
            plan.start(local)
            ngather += 1
    if plan is not None:
        plan.wait()
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tm = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        dt = float(tm.item())
        if rank == 0 and ngather > 0:
            got = plan.result()
            assert got.shape == (per * world, 6, nwn) and all(float(got[r * per, 0, 0]) == r for r in range(world))
    if rank == 0:
        # the stub goes through the same emit() as a measurement, padded with a detail block of the size a real run carries
        emit({"metric": "stub", "value": 0.0, "unit": "evals/s", "n_gpus": world, "n_ranks_seen": seen, "steps": args.steps,
              "warmup": args.warmup, "warmup_steps_run": n_w, "ms_per_step": dt / max(args.steps, 1) * 1e3, "stub": True,
              "gather_every": args.gather_every, "gathers_in_timed_region": ngather, "config": {"workload": "stub: no-op step on CPU tensors (gloo)"},
              "workloads": {f"w{i}": {"value": float(i), "ms_per_step": 1.0, "dtype": "f64", "roofline": {"frac": 0.5, "pad": "x" * 2000}}
                            for i in range(8)}}, args.detail_file)
    if world > 1:
        dist.destroy_process_group()
    return 0


# ------------------------------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="c4")
    ap.add_argument("--profiles-per-gpu", type=int, default=128, help="batch of the shard workloads (c4shard, c2lc, c4brd; c5: a "
                    "quarter of it); c4 / c5full split BASELINE's 1024 / 256 profiles over the GPUs instead")
    ap.add_argument("--min-seconds", type=float, default=1.0, help="timed duration of each secondary workload; also the least "
                    "warm-up duration before the headline's K steps (the shader clock settles during the first second)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-events", action="store_true", help="do not record per-kernel HIP events in the timed region")
    ap.add_argument("--cpu-sample", type=int, default=0, help="profiles of the workload timed on the CPU (0 = ~10-15 s worth)")
    ap.add_argument("--no-extra", action="store_true", help="headline only: skip the c3 / c5 / c2lc / single-profile sub-measurements")
    ap.add_argument("--no-single", action="store_true", help=argparse.SUPPRESS)  # older name of --no-extra
    ap.add_argument("--no-pmc", action="store_true", help="do not collect counters with a rocprofv3 child run")
    ap.add_argument("--save-pmc", default="", help="write the collected per-launch counters (keyed by the csrc hash) to this file")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step from a captured HIP graph (kernel events are then taken in extra untimed steps)")
    ap.add_argument("--real-kind", type=int, default=0, help="8 = dbl build, 4 = sgl build; default: 4 for c5, else 8")
    ap.add_argument("--gather-every", type=int, default=1, help="N > 1: gather the spectral outputs to rank 0 every K-th step only "
                    "(default 1 = every step, the north-star's one gather per pass); separates compute scaling from the collective")
    ap.add_argument("--cpu-procs", type=int, default=16, help="reference processes of the all-cores CPU leg (the GPU's share of the host)")
    ap.add_argument("--detail-file", default="", help="where the full record goes (default gpurun_out/bench_detail.json)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--pmc-workloads", default="", help=argparse.SUPPRESS)
    ap.add_argument("--pmc-manifest", default="", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.pmc_child:
        return pmc_child(args)

    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        return launch_ranks(args.gpus, sys.argv[1:])
    world = int(env_world or "1")
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: WORLD_SIZE={world} but --gpus {args.gpus}: start it with matching values", file=sys.stderr)
        return 2
    if os.environ.get("MONORTM_BENCH_STUB") == "1":
        return stub_rank(args, world, rank)

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    # rehearsal on a one-GPU box: MONORTM_BENCH_BACKEND=gloo lets several ranks share the card (RCCL refuses that)
    backend = os.environ.get("MONORTM_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend != "nccl":
        local = local % ndev
    elif local >= ndev:
        print(f"bench.py: rank {rank} needs device {local} but only {ndev} visible", file=sys.stderr)
        return 2
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from monortm_amd import distributed as D

    res = Resident(args.workload, rank, local, args.profiles_per_gpu, real_kind=args.real_kind, world=world)
    strong = args.workload in ("c4", "c5full")   # BASELINE's batch split over the GPUs; the other workloads are per-GPU shards
    if strong:
        nprof_total = (C4_PROFILES if args.workload == "c4" else 2 * C5_PROFILES)
    else:
        nprof_total = len(res.profs) * world
    # the single RCCL gather of the per-profile spectral outputs (north_star): buffers allocated once, issued
    # asynchronously so that it overlaps the next step's kernels; the last one is waited for inside the timed region
    # (round 5: the kernels write their six spectral outputs into one block, two blocks alternate between steps, and the gather
    # takes the block as it is - the torch.stack + copy that used to feed it cost 15 us per step on the compute stream,
    # profiles/r05_gather_overlap.txt)
    plan = None
    if world > 1:
        res.batch.pingpong = True
        plan = D.GatherPlan(nprof_total, res.batch.spectral_block(), field_major=True)
        # what the collective costs by itself (nothing else on the GPU): 20 gathers of the real block, back to back
        res.batch.step()
        plan.start(res.batch.spectral_block())
        plan.wait()
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(20):
            plan.start(res.batch.spectral_block())
            plan.wait()
        torch.cuda.synchronize()
        gather_us = torch.tensor([(time.perf_counter() - t0) / 20 * 1e6], dtype=torch.float64, device=dev)
        dist.all_reduce(gather_us, op=dist.ReduceOp.MAX)
        gather_us = float(gather_us.item())
    m = timed_steps(torch, dist, res, args.steps, args.warmup, args.min_seconds, plan=plan, graph=args.graph,
                    events=not args.no_events, world=world, gather_every=max(1, args.gather_every))
    dt = m["dt"]
    seen, ordinals = world, [local]
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        etot = torch.tensor([res.e_step, 1.0], dtype=torch.float64, device=dev)
        dist.all_reduce(etot, op=dist.ReduceOp.SUM)
        e_all, seen = float(etot[0].item()), int(round(float(etot[1].item())))
        ords = [None] * world
        dist.all_gather_object(ords, (rank, local, torch.cuda.get_device_properties(local).name))
        ordinals = ords
    else:
        e_all = res.e_step

    rc = 0
    if rank == 0:
        value = e_all * args.steps / dt
        avg_ms = m["kernel_ms"]["lines"]
        out = {
            "metric": "(wavenumber x layer x line) optical-depth evals/sec",
            "value": value,
            "unit": "evals/s",
            "n_gpus": world,
            "n_ranks_seen": seen,
            "rank_devices": ordinals,
            "steps": args.steps,
            "warmup": args.warmup,
            "warmup_steps_run": m["warmup_steps_run"],
            "ms_per_step": dt / args.steps * 1e3,
            "timed_ms": dt * 1e3,      # = steps x ms_per_step: the whole timed region, to hold against the driver's wall clock
            "higher_is_better": True,
            # c4 / c5full: BASELINE's batch (1024 / 256 profiles) is fixed and split over the GPUs; shard workloads keep the per-GPU batch
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f64" if res.real_kind == 8 else "f32",
            "data": "synthetic",
            "config": res.config(world, args.graph),
            "profiles_per_sec": nprof_total * args.steps / dt,
            "kernel_ms_per_step": m["kernel_ms"],
        }
        if world > 1:
            out["gather_every"] = max(1, args.gather_every)
            out["gather_us_alone"] = gather_us
        extra = {}
        if world == 1 and not (args.no_extra or args.no_single) and args.workload == "c4":
            # the other BASELINE configurations that fit one GPU, in the same line: each timed for >= min-seconds
            for name, graph in (("c4shard", False), ("c3", False), ("c5", False), ("c5full", False), ("c2lc", False), ("c4brd", False),
                                ("c2real", False), ("c2", True)):
                try:
                    r2 = Resident(name, 0, local, args.profiles_per_gpu, tmp=res.tmp)
                    m2 = measure_by_duration(torch, r2, args.min_seconds, graph=graph)
                    extra[name] = {"config": r2.config(1, graph), "dtype": "f64" if r2.real_kind == 8 else "f32",
                                   "value": r2.e_step * m2["steps"] / m2["dt"], "unit": "evals/s", "steps": m2["steps"],
                                   "ms_per_step": m2["dt"] / m2["steps"] * 1e3, "timed_ms": m2["dt"] * 1e3,
                                   "profiles_per_sec": len(r2.profs) * m2["steps"] / m2["dt"],
                                   "kernel_ms_per_step": m2["kernel_ms"], "_e_step": r2.e_step}
                    if name == "c3" and not args.no_cpu_baseline:
                        # the reference on a slice of configs[2] in the same run: ONE layer x every third wavenumber of the grid x the
                        # 100000 lines = 3.3e8 evals, ~25 s on one core (no more extrapolation from the 50-channel
                        # case; the rate per eval does not depend on the number of wavenumbers - the whole grid takes 72 s, 1.4e7 evals/s:
                        # half the rate of the 500-line case, the reference's (39, 250000) arrays miss the caches)
                        from monortm_amd import synth as _synth
                        p3 = r2.profs[0]
                        one = _synth.Profile(wn=p3.wn[::3], p=p3.p[:1], t=p3.t[:1], tz=p3.tz[:2], wkl=p3.wkl[:1], wbrodl=p3.wbrodl[:1],
                                             clw=p3.clw[:1], irt=3, dvset=p3.dvset * 3)
                        thin = _synth.Profile(wn=p3.wn[::50], p=p3.p[:1], t=p3.t[:1], tz=p3.tz[:2], wkl=p3.wkl[:1], wbrodl=p3.wbrodl[:1],
                                              clw=p3.clw[:1], irt=3, dvset=p3.dvset * 50)
                        try:
                            extra[name]["cpu_baseline"] = cpu_baseline(r2.rec, [one], 1, census_profile=thin)
                        except Exception as e:
                            extra[name]["cpu_baseline"] = {"value": None, "error": f"{type(e).__name__}: {e}"}
                    r2.close()
                except Exception as e:  # a secondary workload never takes the headline down
                    extra[name] = {"error": f"{type(e).__name__}: {e}"}
        # counters: live child run under rocprofv3 --pmc; else the committed summary if it matches this source tree
        pmc, source = None, None
        names = [args.workload] + [k for k in ("c4shard", "c3", "c5", "c5full", "c2lc", "c4brd", "c2real") if k in extra and "error" not in extra[k]]
        nested = any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "") \
            or "HSA_TOOLS_LIB" in os.environ
        if world == 1 and not args.no_pmc and not nested:
            try:
                t0 = time.perf_counter()
                pmc = collect_pmc(names, args.profiles_per_gpu)
                source = f"live: rocprofv3 --pmc child run of this script ({time.perf_counter() - t0:.0f} s, 3 launches per workload)"
                if args.save_pmc:
                    with open(args.save_pmc, "w") as f:
                        json.dump({"csrc_hash": csrc_hash(), "profiles_per_gpu": args.profiles_per_gpu, "passes": PMC_PASSES,
                                   "per_launch": pmc}, f, indent=1)
            except Exception as e:
                source = f"live collection failed: {type(e).__name__}: {str(e)[:200]}"
        if pmc is None:
            for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_per_launch.json")), reverse=True):
                try:
                    j = json.load(open(f))
                    if j.get("csrc_hash") == csrc_hash() and j.get("profiles_per_gpu") == args.profiles_per_gpu:
                        pmc = j["per_launch"]
                        source = (source + "; " if source else "") + f"{os.path.relpath(f, ROOT)} (csrc hash matches)"
                        break
                except Exception:
                    pass
            else:
                source = (source + "; " if source else "") + "no committed counter summary matches this source tree (stale files are refused)"
        get = lambda w: (pmc or {}).get(w, {}).get("lines_kernel")  # noqa: E731
        out["roofline"] = roofline_from_counters(get(args.workload), avg_ms, res.e_step, source, f32=res.real_kind == 4,
                                                 nwn=res.profs[0].nwn)
        out["roofline"]["launches"] = m["lines_launches"]
        if out["roofline"]["frac"] is not None:
            assert 0.0 < out["roofline"]["frac"] <= 1.0, out["roofline"]
        out["roofline_hbm_model"] = hbm_model(avg_ms, res.e_step)
        for name, x in extra.items():
            if "error" in x:
                continue
            e2 = x.pop("_e_step")
            if name != "c2":
                x["roofline"] = roofline_from_counters(get(name), x["kernel_ms_per_step"]["lines"], e2, source, f32=x["dtype"] == "f32",
                                                       nwn=x["config"]["wavenumbers"])
                if name == "c3":
                    x["roofline"]["note"] = ("kernel family = physics_kernel + far_plan_kernel + far_kernel (every level) + lines_kernel, all between the "
                                             "'lines' events; the counted FP64 work per eval is ~1.4 flop where direct summation needs ~25: far lines are "
                                             "expanded once per interval of wavenumbers (Chebyshev sums in levels) instead of being evaluated per "
                                             "wavenumber, so frac prices far less work than round 4's did at twice the time")
                if pmc and name in pmc:
                    x["finish_kernel_counters"] = {k: v for k, v in pmc[name].get("finish_kernel", {}).items()
                                                   if k in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE", "_dispatches_per_step")}
            else:
                x["launch"] = "hip graph replay"
        if "c4shard" in extra and "error" not in extra["c4shard"]:
            # `value` is configs[3] whole on this one GPU; the 128-profile share one of 8 GPUs holds runs at a lower rate (the
            # fixed cost of a launch is spread over an eighth of the profiles): what the weak-scaling unit delivers
            out["value_weak_shard"] = extra["c4shard"]["value"]
            out["value_weak_shard_what"] = "the 128-profile share of one of 8 GPUs (1024 / 8), evals/s - see workloads.c4shard"
        if extra:
            out["workloads"] = extra
            if "c2" in extra:
                out["configs1_single_profile"] = extra["c2"]
        if world == 1 and not (args.no_extra or args.no_single) and args.workload in ("c4", "c4shard", "c2lc"):
            try:
                out["dropin"] = dropin_latency(res.rec, res.profs[:128], res.tmp)
            except Exception as e:
                out["dropin"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not args.no_cpu_baseline:
            try:
                if args.workload == "c3":
                    out["cpu_baseline"] = {"value": None, "unit": "evals/s", "cores": 1, "kind": "reference",
                                           "sample": "not timed for c3 (about an hour of CPU work); see the c4shard line"}
                else:
                    # ~10-15 s of single-core work: 256 configs[3] profiles, 64 c5 runs (4x the channels), the one c2 profile
                    ns = args.cpu_sample or {"c4": 256, "c4shard": 256, "c2lc": 64, "c5": 64, "c5full": 64}.get(args.workload, 1)
                    sample = res.profs if ns <= len(res.profs) else build_workload(args.workload, 0, ns)[1]
                    out["cpu_baseline"] = cpu_baseline(res.rec, sample, min(ns, len(sample)), sgl=res.real_kind == 4, ncpu_procs=args.cpu_procs)
                    if "c5full" in extra and "error" not in extra["c5full"]:
                        # configs[4] is the reference's "sgl" build: its own CPU leg (32 profiles x 2 views of the same workload)
                        rec5, profs5, _, _, _ = build_workload("c5", 0, 128)
                        extra["c5full"]["cpu_baseline"] = cpu_baseline(rec5, profs5, 64, sgl=True)
            except Exception as e:
                out["cpu_baseline"] = {"value": None, "error": f"{type(e).__name__}: {e}"}
        # evals count the lines that the 25 cm-1 rule rejects (the reference visits them); the kernel never touches
        # those: the rate of line shapes actually evaluated, where the branch census of the CPU leg gives the share
        try:
            cpf = out.get("cpu_baseline", {}).get("census", {}).get("cut_pass_frac")
            if cpf:
                out["shapes_evaluated_per_s"] = out["value"] * cpf
                out["roofline"]["shapes_evaluated_per_s"] = out["value"] * cpf
            for k, x in out.get("workloads", {}).items():
                c2 = x.get("cpu_baseline", {}).get("census", {}).get("cut_pass_frac") if isinstance(x, dict) else None
                if c2:
                    x["shapes_evaluated_per_s"] = x["value"] * c2
        except Exception:
            pass
        emit(out, args.detail_file)
    res.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return rc


if __name__ == "__main__":
    sys.exit(main() or 0)
