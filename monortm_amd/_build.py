"""In-tree builds of the native pieces (hipcc for gfx950, amdflang for the Fortran shim).

The built libraries stay inside the repository (monortm_amd/lib/) so that they travel to the GPU
box with the source snapshot; nothing is installed into site-packages.
"""
from __future__ import annotations

import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
FSRC = os.path.join(PKG, "fortran")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "libmonortm_hip.so")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FC = os.environ.get("MONORTM_FC", "/opt/rocm/bin/amdflang")
HIP_SOURCES = ["api.hip", "lines_kernel.hip", "lines_ms_kernel.hip", "far_kernel.hip", "continuum_kernel.hip", "xsec_kernel.hip", "rtm_kernel.hip", "line_table.cpp"]
HIP_DEPS = HIP_SOURCES + ["device_common.hpp", "lineshape.hpp", "lines_device.hpp", "lines_asm.hpp", "lines_ms_asm.hpp", "line_table.hpp", "tables/monortm_tables.h",
                           "../../include/monortm_hip.h"]
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value", "-Wno-pass-failed",
             "-Wno-unused-const-variable"]


# lines_ms_kernel.hip: the machine-level loop-invariant code motion hoists the materialisation of ~30 FP64 constants (the polynomial
# of the prepare stage's exp) out of the chunk loop; they do not fit beside the 48 fixed registers of the class loops, are spilled, and
# every use inside the prepare stage becomes a scratch load - 7 GB of scratch traffic per configs[3] launch (LABNOTES round 6)
# lines_kernel.hip: the same switch measured on every workload (round 6, one box, interleaved): configs[4] whole 0.952 -> 0.932 ms, its
# 32-profile share 0.149 -> 0.145, c4brd (the IBRD instantiation that needed two spilled registers) 0.209 -> 0.192, configs[2] 2.292 ->
# 2.276, c2lc / c2real / the 128-profile shard within 0.6 % - identical results (the same operations, scheduled differently)
# continuum_kernel.hip: finish_mw_kernel 0.106 -> 0.089 ms on configs[3] whole, 0.107 -> 0.090 on configs[4] whole, 0.0180 -> 0.0166 on the
# 128-profile shard (a latency chain at eight waves per SIMD: fewer live constants, fewer scalar spills).  far_kernel.hip and
# rtm_kernel.hip: no gain / 3 % worse - they keep the default.
_NO_LICM = ["-mllvm", "-disable-machine-licm"]
HIP_FILE_FLAGS = {"lines_ms_kernel.hip": _NO_LICM, "lines_kernel.hip": _NO_LICM, "continuum_kernel.hip": _NO_LICM}


def _check_flags(flags: list[str]) -> None:
    """The shipped library is built without any of the sources' timing-experiment / A-B switches (device_common.hpp refuses them
    too, for builds that do not come through here): HIPCC_FLAGS-style additions from the environment are inspected."""
    extra = os.environ.get("MONORTM_EXTRA_HIPFLAGS", "").split()
    bad = [f for f in flags + extra if f.startswith(("-DFAR_ABL_", "-DMONORTM_ABLATE_", "-DMONORTM_NO_", "-DLINES_TIMING", "-DLINES_CLASS_STATS",
                                                     "-DMW_TIMING"))]
    if bad and os.environ.get("MONORTM_EXPERIMENT") != "1":
        raise RuntimeError(f"experiment switches in the build flags {bad}: set MONORTM_EXPERIMENT=1 (and use tools/build_variant.sh)")


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_hip(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 -> monortm_amd/lib/libmonortm_hip.so (cross-compiles without a GPU).
    One object per translation unit (monortm_amd/lib/obj/), compiled side by side and only when stale, then one link."""
    os.makedirs(LIBDIR, exist_ok=True)
    deps = [os.path.join(CSRC, d) for d in HIP_DEPS]
    headers = [d for d in deps if not d.endswith((".hip", ".cpp"))]
    if not (force or _stale(LIB, deps)):
        return LIB
    if not shutil.which(HIPCC) and not os.path.exists(HIPCC):
        raise RuntimeError(f"hipcc not found ({HIPCC}); the HIP extension cannot be built")
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    cflags = [f for f in HIP_FLAGS if f != "-shared"] + os.environ.get("MONORTM_EXTRA_HIPFLAGS", "").split()
    _check_flags(cflags)
    if os.environ.get("MONORTM_EXPERIMENT") == "1":
        cflags.append("-DMONORTM_EXPERIMENT=1")
    jobs, objs = [], []
    for src in HIP_SOURCES:
        sp, ob = os.path.join(CSRC, src), os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        objs.append(ob)
        if force or _stale(ob, [sp] + headers):
            cmd = [HIPCC, *cflags, *HIP_FILE_FLAGS.get(src, []), "-c", sp, "-o", ob]
            if verbose:
                print(" ".join(cmd))
            jobs.append((cmd, subprocess.Popen(cmd)))
    for cmd, pr in jobs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


def build_fortran_shim(force: bool = False) -> dict:
    """amdflang: the ISO_C_BINDING modules ModmMod / RTMmono (+ CntnmFactors, lblparams) and the dump
    harness linked against them -> monortm_amd/lib/harness_hip_dbl."""
    build_hip(force=False)
    out = os.path.join(LIBDIR, "harness_hip_dbl")
    moddir = os.path.join(LIBDIR, "fmod_dbl")
    srcs = [os.path.join(FSRC, f) for f in ("monortm_hip_c.f90", "lblparams_hip.f90", "cntnmfactors_hip.f90",
                                            "rtmmono_hip.f90", "xsec_hip.f90", "modm_hip.f90")]
    harness = os.path.join(ROOT, "examples", "harness.f90")
    # XSREAD for builds without the reference tree (the harness calls it for IXSECT = 1 cases)
    xsread = os.path.join(FSRC, "xsread_hip.f90")
    srcs = srcs + [xsread]
    if force or _stale(out, srcs + [harness, LIB]):
        os.makedirs(moddir, exist_ok=True)
        dbl = ["-fdefault-integer-8", "-fdefault-real-8"]
        objs = []
        for s in srcs:
            o = os.path.join(moddir, os.path.basename(s)[:-4] + ".o")
            subprocess.check_call([FC, "-c", *dbl, "-O2", "-module-dir", moddir, "-I", moddir, s, "-o", o])
            objs.append(o)
        subprocess.check_call([FC, *dbl, "-O2", "-I", moddir, harness, *objs, "-L", LIBDIR, "-lmonortm_hip",
                               f"-Wl,-rpath,{LIBDIR}", "-o", out])
    # the same harness for a single-precision caller ("sgl" flag set: default REAL = 4 bytes); the shim converts to
    # C doubles, the GPU computes in f64 (>= the reference's precision)
    out_s = os.path.join(LIBDIR, "harness_hip_sgl")
    if force or _stale(out_s, srcs + [harness, LIB]):
        mods = os.path.join(LIBDIR, "fmod_sgl")
        os.makedirs(mods, exist_ok=True)
        objs = []
        for s in srcs:
            o = os.path.join(mods, os.path.basename(s)[:-4] + ".o")
            subprocess.check_call([FC, "-c", "-O2", "-module-dir", mods, "-I", mods, s, "-o", o])
            objs.append(o)
        subprocess.check_call([FC, "-O2", "-I", mods, harness, *objs, "-L", LIBDIR, "-lmonortm_hip",
                               f"-Wl,-rpath,{LIBDIR}", "-o", out_s])
    # the stand-alone IATM=0 driver (MONORTM.IN / MONORTM_PROF.IN / TAPE3 -> MONORTM.OUT), batched C ABI calls
    drv = os.path.join(LIBDIR, "monortm_hip")
    dsrc = [os.path.join(FSRC, f) for f in ("monortm_hip_c.f90", "lblparams_hip.f90", "xsec_hip.f90", "xsread_hip.f90",
                                            "atm_models_data.f90", "lblatm_front.f90", "netcdf3_writer.f90", "monortm_driver.f90")]
    if force or _stale(drv, dsrc + [LIB]):
        dmod = os.path.join(LIBDIR, "fmod_drv")
        os.makedirs(dmod, exist_ok=True)
        # the driver is a "dbl" program like the reference's default build: default REAL = 8 bytes
        subprocess.check_call([FC, "-O2", "-fdefault-real-8", "-module-dir", dmod, "-I", dmod, *dsrc, "-L", LIBDIR,
                               "-lmonortm_hip", f"-Wl,-rpath,{LIBDIR}", "-o", drv])
    return {"harness": out, "harness_sgl": out_s, "moddir": moddir, "driver": drv}
