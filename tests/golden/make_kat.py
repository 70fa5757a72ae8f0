#!/usr/bin/env python3
"""Generate tests/golden/functions/kat_functions.npz: known answers of the small functions on the path (W4, SD_Humlicek, SDVOIGT,
RADFN, AtoB, ODCLW_TKC) from the REFERENCE ITSELF, compiled here by amdflang with the reference's "dbl" flags.

W4 / SD_Humlicek / SDVOIGT are PRIVATE in the reference's ModmMod (src/modm.f90:10), so - as SURVEY.md 8(c)(iii)
prescribes - src/modm.f90 is copied to a scratch directory under /tmp, its PRIVATE line is dropped there, and the
copy is compiled only there.  Nothing of the reference enters the repository: the fixture holds the input grids
(chosen here to straddle every region boundary) and the returned values.

    python tests/golden/make_kat.py        (needs /root/reference; run in the build container)
"""
import os
import shutil
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("MONORTM_REFERENCE", "/root/reference")
FC = "/opt/rocm/bin/amdflang"
DBL = ["-fdefault-integer-8", "-fdefault-real-8", "-O0"]
SCR = "/tmp/monortm_kat"
KEYS = ("w4", "sdh", "sdv", "radfn", "atob", "tkc", "tips", "hwd", "bbfn", "intens", "hwc", "lortz", "sdvlsf")
WIDE = ("intens", "hwc", "lortz", "sdvlsf")   # 12 arguments per row
ISONM = (6, 9, 18, 5, 6, 3, 3, 3, 2, 1, 2, 1, 3, 1, 2, 2, 1, 2, 5, 3, 2, 1, 3, 2, 1, 2, 1, 1, 1, 1, 3, 1, 1, 1, 2, 1, 2, 2, 1)  # src/tips_2003.f90:361-369


def nextafter_set(v, k=2):
    out = [v]
    lo = hi = v
    for _ in range(k):
        lo = np.nextafter(lo, -np.inf)
        hi = np.nextafter(hi, np.inf)
        out += [lo, hi]
    return out


def inputs():
    rng = np.random.default_rng(20261004)
    g = {}
    # W4: S = |x| + y across 15 and 5.5, y against 0.195|x| - 0.176, both signs of x, exact boundary values +- 2 ulp
    pts = []
    for x in (-12.0, -4.0, -1.5, -0.3, 0.0, 0.2, 0.9, 1.0, 2.5, 3.0, 5.0, 8.0, 14.0, 20.0):
        for s in (15.0, 5.5):
            for y in nextafter_set(s - abs(x)):
                if y >= 0:
                    pts.append((x, y))
        for y in nextafter_set(0.195 * abs(x) - 0.176):
            if y >= 0 and abs(x) + y < 5.5:
                pts.append((x, y))
        for y in (0.0, 1e-6, 1e-3, 0.05, 0.3, 1.0, 4.0, 10.0, 30.0, 1000.0):
            pts.append((x, y))
    pts += [(float(a), float(b)) for a, b in zip(rng.uniform(-20, 20, 200), 10 ** rng.uniform(-6, 1.5, 200))]
    w = np.array(pts)
    g["w4"] = np.column_stack([w, np.zeros((len(w), 2))])
    # SD_Humlicek: two points per call, II/III boundary at S = 6 (not 5.5), the larger region wins
    q = []
    for x1, y1 in ((0.1, 0.05), (1.0, 0.01), (2.0, 3.0), (3.0, 3.0), (5.0, 0.9), (7.0, 8.0), (-0.5, 0.2), (0.0, 6.0), (10.0, 5.0)):
        for x2, y2 in ((0.1, 0.3), (0.4, 5.6), (1.0, 5.0), (2.0, 13.0), (-3.0, 0.2), (20.0, 1.0)):
            q.append((x1, y1, x2, y2))
    for s in (15.0, 6.0):
        for y in nextafter_set(s - 1.25):
            q.append((1.25, y, 0.3, 0.1))
            q.append((0.3, 0.1, -1.25, y))
    for y in nextafter_set(0.195 * 2.0 - 0.176):
        q.append((2.0, y, 0.1, 4.0))
    g["sdh"] = np.array(q)
    # SDVOIGT(deltnu, alphal, alphad, sdep): Voigt (|sdep| <= 1e-4) and speed-dependent; zeta == 1 shortcut (alphad = 0)
    v = []
    for dn in (0.0, 1e-6, 3e-5, 1e-4, 2e-3, 0.05, 1.0, 25.0, -4e-5):
        for al, ad in ((2e-7, 2e-6), (5e-6, 2e-6), (1e-4, 1e-4), (3e-3, 4e-5), (0.05, 1e-4), (0.05, 0.0)):
            for sd in (0.0, 5e-5, 0.08, 0.15):
                if ad == 0.0 and sd > 1e-4:
                    continue
                v.append((dn, al, ad, sd))
    g["sdv"] = np.array(v)
    # RADFN(vi, xkt): x = vi/xkt <= 0.01, <= 10, > 10, xkt <= 0; exact thresholds +- 2 ulp
    r = []
    for xkt in (150.0, 205.87, 0.0, -1.0):
        for vi in (1e-3, 0.5, 2.0, 30.0, 1000.0, 2500.0, 30000.0):
            r.append((vi, xkt))
    for xkt in (100.0, 208.5):
        for t in (0.01, 10.0):
            for vi in nextafter_set(t * xkt):
                r.append((vi, xkt))
    g["radfn"] = np.column_stack([np.array(r), np.zeros((len(r), 2))])
    # AtoB on the TIPS grid (60 + 25 k, 119 points): first / last intervals (3-point branch), interior, exact nodes, limits
    tab = 1.0 + 0.02 * (60.0 + 25.0 * np.arange(119)) ** 1.5 * (1 + 0.1 * np.sin(np.arange(119) / 7.0))
    aa = [70.0, 84.9, 85.0, 85.1, 100.0, 110.0, 135.0, 216.7, 250.0, 296.0, 300.0, 1234.5, 2960.0, 2984.9, 2985.0, 2985.1, 2999.0, 3000.0, 3010.0]
    aa += list(rng.uniform(70, 3000, 40))
    g["atob_tab"] = tab
    g["atob"] = np.column_stack([np.array(aa), np.zeros((len(aa), 3))])
    # ODCLW_TKC(wn, temp, clw): 0.5-500 GHz, -40..50 C
    c = [(wn, t, clw) for wn in (0.02, 0.3, 0.79, 1.0, 3.0, 6.5, 16.0) for t in (233.15, 255.0, 273.15, 285.0, 300.0) for clw in (0.0, 0.013, 0.05)]
    g["tkc"] = np.column_stack([np.array(c), np.zeros((len(c), 1))])
    # TIPS_2003(39, T, scor): every (molecule, isotopologue <= 9) at the limits, on grid nodes, in the 3-point end intervals and
    # inside; includes molecule 34 (Q = 1), molecule 39 (the reference's stale-QT path: scor = 1) and the slots TIPS never writes
    tt = [70.0, 75.0, 84.999, 110.0, 216.7, 250.0, 296.0, 310.0, 1200.0, 2985.0, 2999.0, 3000.0]
    g["tips"] = np.array([(t, mol, iso, 0.0) for t in tt for mol in range(1, 40) for iso in range(1, 10)])
    # HALFWHM_D(mol, iso, xnu, T): EVERY (molecule, isotopologue) that carries a mass (98 slots, src/isotope.incl:51-167) at two
    # temperatures - the Doppler width is where the product's generated mass table enters, and this is the reference's own
    # COMMON /ISVECT/ answering
    g["hwd"] = np.array([(m, i, v, t) for m in range(1, 40) for i in range(1, min(9, ISONM[m - 1]) + 1)
                         for v, t in ((22.235, 216.7), (1604.0, 296.0))], float)
    # bb_fn(v, fbeta): microwave to UV, 2.75 K (overflow of exp -> 0) to 330 K
    bb = [(v, 1.4387752 / t) for v in (0.3, 1.9835, 22.235, 183.31, 667.0, 2500.0, 15798.0, 57800.0) for t in (2.75, 77.0, 216.7, 288.2, 330.0)]
    g["bbfn"] = np.column_stack([np.array(bb), np.zeros((len(bb), 2))])
    radct = 6.62606876E-27 * 2.99792458E+10 / 1.3806503E-16
    # INTENS(T, S0s, Es, RADCT, T0, Xnus, XIPSF)
    rows = []
    for t in (70.0, 200.0, 216.7, 296.0, 320.0):
        for xnu, es, s0 in ((0.7417, 0.0, 3e-25), (22.235, 446.51, 1.3e-24), (60.3, 1874.2, 7e-26), (667.38, 2338.7, 4e-20), (2349.1, 106.1, 3e-18)):
            rows.append((t, s0, es, radct, 296.0, xnu, 0.5 + 0.003 * t))
    g["intens"] = np.array([r + (0.0,) * (12 - len(r)) for r in rows])
    # HALFWHM_C(AF, AS, RT, XTILD, RHORAT, MOL, rho_molec(MOL)): incl. the H2O self width 0 -> 5 AF rule
    rows = []
    for mol, af, asw in ((1, 0.0803, 0.41), (1, 0.0803, 0.0), (2, 0.07, 0.09), (3, 0.069, 0.095), (7, 0.0452, 0.0441), (4, 0.075, 0.1)):
        for rt, rho in ((216.7 / 296.0, 0.03), (1.0, 1.0), (310.0 / 296.0, 0.5)):
            for n in (0.45, 0.76):
                rows.append((af, asw, rt, n, rho, mol, rho * {1: 0.012, 2: 4e-4, 3: 3e-7, 7: 0.209, 4: 3.2e-7}[mol]))
    g["hwc"] = np.array([r + (0.0,) * (12 - len(r)) for r in rows])
    # LSF_LORTZ(XF, RP, RP2, AIP, BIP, HWHM, WN, Xnu, MOL): every branch - generic / O2 / CO2, coupled (-1, -3, -5) and not, both
    # sides of the 25 cm-1 rule and of WN + Xnu = 25, the exact thresholds
    rows = []
    for mol in (1, 3, 7, 2):
        for xf, aip, bip in ((0.0, 0.0, 0.0), (-1.0, 0.12, -0.01), (-3.0, 0.2, 0.015), (-5.0, -0.08, 0.004)):
            for wn, xnu in ((2.0, 1.9835), (2.0, 22.9), (2.0, 23.0), (2.0, 23.1), (1.9835, 1.9835), (5.0, 30.0), (5.0, 30.1), (40.0, 15.0), (40.0, 14.9),
                            (0.5, 24.5), (20.0, 45.0 + 1e-9)):
                for rp, hw in ((1.0, 0.08), (0.02, 0.0016)):
                    rows.append((xf, rp, rp * rp, aip, bip, hw, wn, xnu, mol))
    g["lortz"] = np.array([r + (0.0,) * (12 - len(r)) for r in rows])
    # LSF_SDVOIGT(XF, RP, RP2, AIP, BIP, HWHM, WN, Xnu, AD, MOL, SDEP): thin layers, near and far from the centre, with and
    # without speed dependence, the same molecule / coupling matrix (incl. the literal XF.NE.-5 condition for CO2, :659)
    rows = []
    for mol in (1, 3, 7, 2):
        for xf, aip, bip in ((0.0, 0.0, 0.0), (-1.0, 0.12, -0.01), (-3.0, 0.2, 0.015), (-5.0, -0.08, 0.004)):
            for wn, xnu in ((1.9835, 1.9835), (1.98353, 1.9835), (1.9839, 1.9835), (3.0, 1.9835), (24.0, 1.9835), (1.9835, 27.0)):
                for hw, ad, sd in ((4e-6, 2.2e-6, 0.0), (8e-5, 2.2e-6, 0.0), (2e-5, 2.2e-6, 0.11)):
                    rows.append((xf, 0.004, 1.6e-5, aip, bip, hw, wn, xnu, ad, mol, sd))
    g["sdvlsf"] = np.array([r + (0.0,) * (12 - len(r)) for r in rows])
    return g


def main():
    if not os.path.isdir(os.path.join(REF, "src")):
        sys.exit("reference tree not found")
    shutil.rmtree(SCR, ignore_errors=True)
    os.makedirs(SCR)
    src = open(os.path.join(REF, "src", "modm.f90")).read().split("\n")
    kept = [ln for ln in src if ln.strip().upper() != "PRIVATE"]
    assert len(kept) == len(src) - 1, "expected exactly one PRIVATE line in modm.f90"
    open(os.path.join(SCR, "modm_public.f90"), "w").write("\n".join(kept))   # scratch only
    rsrc = open(os.path.join(REF, "src", "RTMmono.f90")).read().split("\n")       # bb_fn is PRIVATE in RTMmono (:3): same treatment
    rkept = [ln for ln in rsrc if not ln.strip().upper().startswith("PRIVATE ::")]
    assert len(rkept) == len(rsrc) - 1, "expected exactly one PRIVATE :: line in RTMmono.f90"
    open(os.path.join(SCR, "rtmmono_public.f90"), "w").write("\n".join(rkept))  # scratch only
    # every other unit comes from the oracle recipe's scratch objects (oracle/Makefile, -O0 "dbl" variant): MODM references
    # the cross-section and LBLATM units, so the whole reference is linked, with only modm.o replaced by the PUBLIC copy
    root = os.path.dirname(os.path.dirname(HERE))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "oracle"), "ref"])
    ref_objs = os.environ.get("MONORTM_REF_SCRATCH", "/tmp/monortm_ref_scratch") + "/dbl_O0"
    objs = sorted(os.path.join(ref_objs, f) for f in os.listdir(ref_objs) if f.endswith(".o") and f not in ("modm.o", "RTMmono.o"))
    # (module files of the scratch copies land in SCR, which comes first on the include path)
    subprocess.check_call([FC, "-c", *DBL, "-I", ".", "-I", os.path.join(REF, "src"), "-I", ref_objs, "rtmmono_public.f90", "-o", "rtmmono_public.o"], cwd=SCR)
    subprocess.check_call([FC, "-c", *DBL, "-I", ".", "-I", os.path.join(REF, "src"), "-I", ref_objs, "modm_public.f90", "-o", "modm_public.o"], cwd=SCR)
    subprocess.check_call([FC, *DBL, "-I", ".", "-I", ref_objs, os.path.join(HERE, "kat_driver.f90"), *objs, "modm_public.o", "rtmmono_public.o", "-o", "kat"], cwd=SCR)
    g = inputs()
    with open(os.path.join(SCR, "kat_in.bin"), "wb") as f:
        for key in KEYS:
            a = np.ascontiguousarray(g[key], np.float64)
            f.write(np.array([len(a)], np.float64).tobytes())
            if key == "atob":
                f.write(np.ascontiguousarray(g["atob_tab"], np.float64).tobytes())
            f.write(a.tobytes())
    subprocess.check_call(["./kat"], cwd=SCR)
    out = np.fromfile(os.path.join(SCR, "kat_out.bin"), np.float64).reshape(-1, 2)
    pos = 0
    res = {}
    for key in KEYS:
        n = len(g[key])
        res[key + "_in"] = g[key]
        res[key + "_out"] = out[pos:pos + n].copy()
        pos += n
    assert pos == len(out)
    res["atob_tab"] = g["atob_tab"]
    np.savez_compressed(os.path.join(HERE, "functions", "kat_functions.npz"), **res)
    print({k: v.shape for k, v in res.items()})


if __name__ == "__main__":
    main()
