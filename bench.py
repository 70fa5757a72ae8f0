#!/usr/bin/env python3
"""Benchmark of the MODM + CALCTMR + RTM hot path on MI355X (BASELINE.json metric:
(wavenumber x layer x line) optical-depth evaluations per second; profiles per second).

    python bench.py --gpus N --steps K --warmup W [--workload c4shard|c2|c3|c5] [--real-kind 8|4]

One process per GPU (launched by torch.distributed.run for N > 1).  A "step" is one pass of the hot
path over the rank's resident batch of profiles: lines kernel, continuum/cloud/total kernel, rtm
kernel (+ for N > 1 the single RCCL gather of the spectral outputs to rank 0).  Inputs are resident in
HBM before the timed region starts.  Rank 0 prints ONE JSON line.

Workloads (synthetic, seeded):
  c4shard  default: BASELINE configs[1] profile (64 layers x 50 channels x 500 lines, f64) batched as
           configs[3] prescribes - 128 sonde-like profiles per GPU (1024 / 8), weak scaling
  c2       configs[1] literally: ONE profile per step (launch-latency bound, reported for reference)
  c3       configs[2]: 1 profile x 64 layers x 10000-wavenumber grid x 100000 lines
  c5       configs[4]: up- and downwelling views with a liquid-water cloud layer, 256 / 8 = 32 profiles per GPU x 200
           channels, single precision (real_kind 4: the reference's "sgl" build)
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BYTES_PER_EVAL = 44.0     # VNU f64 + 9 x 4-byte fields of a TAPE3 line record
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
FP64_PEAK_TFLOPS = 78.6   # vector FP64
FLOPS_PER_EVAL = 40.0     # flop-equivalents of one prepared Lorentz evaluation


def build_workload(name: str, rank: int, per_gpu: int):
    from monortm_amd import synth

    if name == "c4shard":
        rec = synth.synthetic_lines(500)
        wn = synth.c2_channels(50)
        profs = [synth.perturbed_profile(rank * per_gpu + i, wn, nlay=64) for i in range(per_gpu)]
        desc = (f"configs[1] profile (64 layers x 50 channels x 500 lines, f64) batched per configs[3]: "
                f"{per_gpu} profiles per GPU")
    elif name == "c2":
        rec = synth.synthetic_lines(500)
        profs = [synth.c2_profile()]
        desc = "configs[1]: 1 profile x 64 layers x 50 channels x 500 lines, f64"
    elif name == "c3":
        rec = synth.synthetic_lines(100000, seed=20261004)
        a = synth.standard_atmosphere(64)
        wn = 0.5 + 0.005 * np.arange(10000)
        profs = [synth.Profile(wn=wn, p=a["p"], t=a["t"], tz=a["tz"], wkl=a["wkl"], wbrodl=a["wbrodl"], clw=a["clw"],
                               irt=3, dvset=0.005)]
        desc = "configs[2]: 1 profile x 64 layers x 10000-wavenumber grid (0.5-50.495 cm-1) x 100000 lines, f64"
    elif name == "c5":
        rec = synth.synthetic_lines(500)
        wn = synth.c2_channels(200)
        per = 32 if per_gpu == 128 else per_gpu
        profs = [synth.perturbed_profile(rank * per + i, wn, nlay=64, cloud=True, irt=(1 if i % 2 == 0 else 3)) for i in range(per)]
        desc = (f"configs[4]: upwelling + downwelling with a cloud liquid layer, {per} profiles per GPU (256 / 8) x 64 layers x "
                f"200 channels x 500 lines, single precision")
    else:
        raise SystemExit(f"unknown workload {name}")
    return rec, profs, desc


def evals_per_step(rt, profs) -> float:
    """E = sum_profiles NWN * sum_layers NL_layer, NL = physical line records of molecules with a
    non-zero column in the layer."""
    counts = np.array([rt.line_count(m) for m in range(1, profs[0].nmol + 1)], np.float64)
    e = 0.0
    for p in profs:
        e += p.nwn * float(((p.wkl != 0.0) * counts[None, :]).sum())
    return e


def cpu_baseline(rec, profs, nsample: int):
    """Time the reference itself (oracle/_ref/harness_ref_dbl_fast: the reference's own sources compiled by
    amdflang, hot-path units at -O2) on a bounded sample of the same workload, 1 host core (the reference
    is serial).  Falls back to the C restatement (kind "port") if the prebuilt binary is absent."""
    from monortm_amd import caseio, tape3

    sample = profs[:nsample]
    census = None
    counts = {}
    r = np.asarray(rec.mol[rec.iflg >= 0]) % 100
    for m in np.unique(r):
        counts[int(m)] = int((r == m).sum())
    ev = sum(p.nwn * sum(counts.get(m + 1, 0) * int((p.wkl[:, m] != 0).sum()) for m in range(p.nmol)) for p in sample)
    harness = os.path.join(ROOT, "oracle", "_ref", "harness_ref_dbl_fast")
    with tempfile.TemporaryDirectory() as d:
        tp, cp, op = (os.path.join(d, n) for n in ("TAPE3", "case.bin", "out.bin"))
        tape3.write_tape3(tp, rec)
        try:  # cut-pass fraction and Lorentz / Voigt split, counted by the C restatement on one profile
            from oracle.pyoracle import Oracle

            orc = Oracle(tp, sample[0].wn[0], sample[0].wn[-1])
            orc.census(reset=True)
            orc.run(sample[0])
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
            r = subprocess.run([harness, cp, tp, op], cwd=d, capture_output=True, text=True)
            wall = time.perf_counter() - t0
            secs = None
            for line in r.stdout.splitlines():
                if line.startswith("HARNESS_SECONDS"):
                    secs = float(line.split()[1])
            if r.returncode == 0 and secs:
                allc = None
                try:  # the same sample split over every host core, one reference process per core (the program is serial)
                    nc = max(1, min(len(sample), len(os.sched_getaffinity(0))))
                    if nc > 1:
                        procs = []
                        t1 = time.perf_counter()
                        for k in range(nc):
                            part = sample[k::nc]
                            ck, ok = os.path.join(d, f"case{k}.bin"), os.path.join(d, f"out{k}.bin")
                            caseio.write_case(ck, part)
                            procs.append(subprocess.Popen([harness, ck, tp, ok], cwd=d, stdout=subprocess.DEVNULL,
                                                          stderr=subprocess.DEVNULL))
                        rcs = [p.wait() for p in procs]
                        w_all = time.perf_counter() - t1
                        if all(rc == 0 for rc in rcs):
                            allc = {"value": ev / w_all, "unit": "evals/s", "cores": nc,
                                    "note": f"{nc} reference processes side by side, wall {w_all:.2f} s incl. start-up and TAPE3 load"}
                except Exception as e:
                    allc = {"error": str(e)}
                return {"value": ev / secs, "unit": "evals/s", "cores": 1, "kind": "reference", "census": census, "all_cores": allc,
                        "sample": f"{len(sample)} profile(s) of the workload = {ev:.3g} evals in {secs:.2f} s "
                                  f"(MODM+CALCTMR+RTM inside the reference, wall {wall:.2f} s; amdflang, hot path -O2)"}
        from oracle.pyoracle import Oracle

        orc = Oracle(tp, sample[0].wn[0], sample[0].wn[-1])
        t0 = time.perf_counter()
        for p in sample:
            orc.run(p)
        secs = time.perf_counter() - t0
        return {"value": ev / secs, "unit": "evals/s", "cores": 1, "kind": "port", "census": census,
                "sample": f"{len(sample)} profile(s) = {ev:.3g} evals in {secs:.2f} s (oracle/monortm_oracle.c, gcc -O2)"}


def single_profile_line(api, tape3, tmp, local, dev, torch, steps: int = 200):
    """BASELINE configs[1] taken literally - ONE profile per step (64 layers x 50 channels x 500 lines): 1.6e6 evals per
    step cannot fill 256 CUs, the step is bound by the latency of three dependent launches.  Reported beside the
    batched headline so that both readings of configs[1] are on record."""
    rec, profs, desc = build_workload("c2", 0, 1)
    t3 = os.path.join(tmp, "TAPE3_c2")
    tape3.write_tape3(t3, rec)
    rt = api.MonoRTM(t3, profs[0].wn[0], profs[0].wn[-1], device=local)
    b = api.DeviceBatch(rt, profs, device=dev)
    e = evals_per_step(rt, profs)
    b.capture()
    for _ in range(10):
        b.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        b.replay()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    b.check()
    rt.close()
    return {"workload": desc, "value": e * steps / dt, "unit": "evals/s", "ms_per_step": dt / steps * 1e3, "steps": steps,
            "launch": "hip graph replay", "profiles_per_sec": steps / dt}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="c4shard")
    ap.add_argument("--profiles-per-gpu", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-events", action="store_true", help="do not record per-kernel HIP events in the timed region")
    ap.add_argument("--cpu-sample", type=int, default=0, help="profiles of the workload timed on the CPU (0 = ~10-15 s worth)")
    ap.add_argument("--no-single", action="store_true", help="skip the extra configs[1] single-profile measurement")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step from a captured HIP graph (kernel events are then taken in extra untimed steps)")
    ap.add_argument("--real-kind", type=int, default=0, help="8 = dbl build, 4 = sgl build; default: 4 for c5, else 8")
    args = ap.parse_args()
    real_kind = args.real_kind or (4 if args.workload == "c5" else 8)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    # rehearsal on a one-GPU box: MONORTM_BENCH_BACKEND=gloo lets several ranks share the card (RCCL refuses that)
    backend = os.environ.get("MONORTM_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from monortm_amd import api, tape3

    rec, profs, desc = build_workload(args.workload, rank, args.profiles_per_gpu)
    tmp = tempfile.mkdtemp(prefix=f"monortm_bench_r{rank}_")
    t3 = os.path.join(tmp, "TAPE3")
    tape3.write_tape3(t3, rec)
    rt = api.MonoRTM(t3, profs[0].wn[0], profs[0].wn[-1], device=local, real_kind=real_kind)
    batch = api.DeviceBatch(rt, profs, device=dev)
    e_step = evals_per_step(rt, profs)

    from monortm_amd import distributed as D

    nprof_total = len(profs) * world

    if args.graph:
        batch.capture()

    # the single RCCL gather of the per-profile spectral outputs (north_star): buffers allocated once, issued
    # asynchronously so that it overlaps the next step's kernels; the last one is waited for inside the timed region
    plan = D.GatherPlan(nprof_total, batch.spectral_outputs()) if world > 1 else None

    def step():
        if args.graph:
            batch.replay()
        else:
            batch.step()
        if plan is not None:
            plan.start(batch.spectral_outputs())

    for _ in range(args.warmup):
        step()
    if plan is not None:
        plan.wait()
    batch.check()
    torch.cuda.synchronize()
    # events around the dominant (lines) kernel only inside the timed region (not possible inside a graph replay)
    rt.profile(0 if (args.no_events or args.graph) else 1)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
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
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        etot = torch.tensor([e_step], dtype=torch.float64, device=dev)
        dist.all_reduce(etot, op=dist.ReduceOp.SUM)
        e_all = float(etot.item())
    else:
        e_all = e_step

    ms_lines, n_lines = rt.kernel_time(0)
    # the two small kernels are timed in a few extra (untimed) steps so that their events do not sit in the timed region
    rt.profile(7 if (args.graph or args.no_events) else 6)
    for _ in range(5):
        batch.step()
    torch.cuda.synchronize()
    rt.profile(0)
    ms_fin, n_fin = rt.kernel_time(1)
    ms_rtm, n_rtm = rt.kernel_time(2)
    if args.graph or args.no_events:
        ms_lines, n_lines = rt.kernel_time(0)

    if rank == 0:
        value = e_all * args.steps / dt
        avg_ms = ms_lines / max(n_lines, 1)
        ach = BYTES_PER_EVAL * e_step / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(args.workload, {}).get("lines_kernel_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "(wavenumber x layer x line) optical-depth evals/sec",
            "value": value,
            "unit": "evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64" if real_kind == 8 else "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: {desc}", "profiles_per_gpu": len(profs), "layers": profs[0].nlay,
                       "wavenumbers": profs[0].nwn, "lines": int(rt.line_count(0)), "nmol": profs[0].nmol,
                       "evals_per_step_per_gpu": e_step, "launch": "hip graph replay" if args.graph else "3 stream launches",
                       "parallelism": f"profile-sharded x{world}" + (", one RCCL gather/step" if world > 1 else "")},
            "profiles_per_sec": len(profs) * world * args.steps / dt,
            "roofline": {"bound": "hbm", "kernel": "lines_kernel", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "avg_launch_ms": avg_ms, "launches": n_lines,
                         "algorithmic_bytes_per_launch": BYTES_PER_EVAL * e_step},
            "kernel_ms_per_step": {"lines": ms_lines / max(n_lines, 1), "continuum_cloud_total": ms_fin / max(n_fin, 1),
                                   "rtm": ms_rtm / max(n_rtm, 1)},
        }
        # the bound that actually holds: FP64 vector ALU.  SURVEY.md 8(d) prices a prepared Lorentz
        # evaluation at ~40 flop-equivalents; lines cut by the 25 cm-1 window are counted as evals but cost nothing,
        # so this is an upper estimate of the arithmetic rate
        tf = FLOPS_PER_EVAL * e_step / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        busy = None
        try:
            busy = json.load(open(tpath)).get(args.workload, {}).get("lines_kernel_valu_busy")
        except Exception:
            pass
        dense = args.workload == "c3"  # far-field moments replace most evaluations: the per-eval flop model does not apply
        out["roofline_fp64"] = {"bound": "valu_fp64", "achieved": None if dense else tf, "peak": FP64_PEAK_TFLOPS,
                                "unit": "TFLOP/s", "frac": None if dense else tf / FP64_PEAK_TFLOPS,
                                "model_flops_per_eval": FLOPS_PER_EVAL, "valu_busy_pmc": busy,
                                "note": "model: every counted eval costs 40 flop (SURVEY.md 8(d)); not given for c3, where ~27 % of "
                                        "the counted (wn, line) pairs fall outside the 25 cm-1 window and ~70 % of the rest are "
                                        "served by the far-field moments of a tile (DESIGN.md 3.1)"}
        if world == 1 and args.workload == "c4shard" and not args.no_single:
            out["configs1_single_profile"] = single_profile_line(api, tape3, tmp, local, dev, torch)
        if world == 1 and not args.no_cpu_baseline:
            if args.workload == "c3":
                out["cpu_baseline"] = {"value": None, "unit": "evals/s", "cores": 1, "kind": "reference",
                                       "sample": "not timed for c3 (about an hour of CPU work); see the c4shard line"}
            else:
                # ~10-15 s of single-core work: 256 c4shard profiles, 64 c5 profiles (4x the channels), the one c2 profile
                ns = args.cpu_sample or {"c4shard": 256, "c5": 64}.get(args.workload, 1)
                sample = profs if ns <= len(profs) else build_workload(args.workload, 0, ns)[1]
                out["cpu_baseline"] = cpu_baseline(rec, sample, min(ns, len(sample)))
                if real_kind == 4:
                    out["cpu_baseline"]["sample"] += "; the CPU leg is the dbl build (the sgl build is only compiled at -O0 here)"
        print(json.dumps(out))
    rt.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
