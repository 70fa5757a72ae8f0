#!/usr/bin/env python3
"""Generate the golden fixtures: inputs + full-precision outputs of the REFERENCE ITSELF.

Runs only in the build container (needs oracle/_ref/harness_ref_dbl, i.e. the reference
compiled by oracle/Makefile from /root/reference).  Each fixture is one compressed .npz
holding the exact TAPE3 bytes, the harness case bytes and the reference's outputs, so the
tests can replay it anywhere without the reference.

    python tests/golden/make_golden.py            # (re)generate every fixture
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from monortm_amd import caseio, synth, tape3  # noqa: E402
from monortm_amd.tape3 import LineRecords  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
HARNESS = os.path.join(ROOT, "oracle", "_ref", "harness_ref_dbl")

BOLTZ, CLIGHT, AVOGAD = 1.3806503E-16, 2.99792458E+10, 6.02214199E+23
MASS = {1: 18.01, 2: 43.99, 3: 47.98, 4: 44.00, 7: 31.99, 22: 28.01}


def alpha_d(nu, T, mol):
    return (nu / CLIGHT) * np.sqrt(2 * np.log(2) * BOLTZ * T / (MASS[mol] / AVOGAD))


def deep_atmosphere(nlay=16, ptop=0.004, nmol=7):
    """surface -> ~80 km: pressures log-spaced so that alpha_L/alpha_D sweeps 1e3 .. 1e-3
    (all four Humlicek regions)."""
    plev = np.exp(np.linspace(np.log(1013.0), np.log(ptop), nlay + 1))
    z = -7.0 * np.log(plev / 1013.0)
    tlev = np.where(z < 11, 288.2 - 6.5 * z, np.where(z < 25, 216.7, np.minimum(216.7 + 2.2 * (z - 25), 270.0)))
    tlev = np.where(z > 50, np.maximum(270.0 - 2.5 * (z - 50), 190.0), tlev)
    p = np.sqrt(plev[:-1] * plev[1:])
    t = 0.5 * (tlev[:-1] + tlev[1:])
    zmid = 0.5 * (z[:-1] + z[1:])
    air = 2.1e25 * (plev[:-1] - plev[1:]) / 1013.0
    vmr = np.zeros((nlay, max(nmol, 7)))
    vmr[:, 0] = 0.008 * np.exp(-zmid / 2.0) + 5e-6
    vmr[:, 1] = 4e-4
    vmr[:, 2] = 3e-7 * (1 + zmid / 5.0) * np.exp(-np.maximum(zmid - 35, 0) / 8)
    vmr[:, 3] = 3.2e-7
    vmr[:, 4] = 1.5e-7
    vmr[:, 5] = 1.7e-6
    vmr[:, 6] = 0.209
    if nmol >= 22:
        vmr[:, 21] = 0.781
    wkl = vmr[:, :nmol] * air[:, None]
    wbrodl = (0.781 if nmol < 22 else 0.0093) * air
    return dict(p=p, t=t, tz=tlev, wkl=wkl, wbrodl=wbrodl, clw=np.zeros(nlay))


def rec_from(rows):
    """rows: list of dicts with keys vnu,s,alfa,hwhm,epp,n,shift,mol,iso,iflg,sdep,lc (optional
    list of (Y[4],G[4]) records)."""
    cols = {k: [] for k in ("vnu", "sp", "alfa", "epp", "mol", "hwhm", "tmpalf", "pshift", "iflg", "sdep")}
    brd_flg, brd_dat = [], []
    for r in rows:
        v = r["vnu"]
        sp = r["s"] / (v * (1.0 - np.exp(-synth.RADCN2 * v / 296.0)))
        cols["vnu"].append(v); cols["sp"].append(sp); cols["alfa"].append(r["alfa"]); cols["epp"].append(r["epp"])
        cols["mol"].append(r["mol"] + 100 * r.get("iso", 1)); cols["hwhm"].append(r["hwhm"])
        cols["tmpalf"].append(r["n"]); cols["pshift"].append(r["shift"]); cols["iflg"].append(r.get("iflg", 0))
        cols["sdep"].append(r.get("sdep", 0.0))
        brd_flg.append(r.get("brd_flg", [0] * 7)); brd_dat.append(r.get("brd_dat", [0.0] * 21))
        for (y, g) in r.get("lc", []):
            cols["vnu"].append(y[0]); cols["sp"].append(g[0]); cols["alfa"].append(y[1]); cols["epp"].append(g[1])
            cols["mol"].append(int(np.float32(y[2]).view(np.int32))); cols["hwhm"].append(g[2])
            cols["tmpalf"].append(y[3]); cols["pshift"].append(g[3]); cols["iflg"].append(-r["iflg"])
            cols["sdep"].append(0.0)
            brd_flg.append([0] * 7); brd_dat.append([0.0] * 21)
    return LineRecords(**{k: np.asarray(v) for k, v in cols.items()}, brd_flg=np.asarray(brd_flg),
                       brd_dat=np.asarray(brd_dat))


def run_reference(rec: LineRecords, profiles, split=None, xs_dir=None):
    with tempfile.TemporaryDirectory() as d:
        tp, cp, op = (os.path.join(d, n) for n in ("TAPE3", "case.bin", "out.bin"))
        tape3.write_tape3(tp, rec, split_blocks_at=split)
        caseio.write_case(cp, profiles)
        if xs_dir:   # FSCDXS and the xs files are opened by name in the working directory (src/monortm_sub.F90:1341,:1662)
            for f in os.listdir(xs_dir):
                shutil.copy(os.path.join(xs_dir, f), d)
        def big_stack():  # MONORTM_XSEC_SUB / convolve hold hundreds of MB of local arrays (src/monortm_sub.F90:1615-1616, :1758)
            import resource
            resource.setrlimit(resource.RLIMIT_STACK, (resource.RLIM_INFINITY, resource.RLIM_INFINITY))
        r = subprocess.run([HARNESS, cp, tp, op], cwd=d, capture_output=True, text=True, preexec_fn=big_stack)
        if r.returncode != 0 or "HARNESS_SECONDS" not in r.stdout:
            raise RuntimeError(f"reference harness failed: rc={r.returncode}\n{r.stdout[-2000:]}\n{r.stderr[-2000:]}")
        dumps = caseio.read_dump(op)
        return open(tp, "rb").read(), open(cp, "rb").read(), dumps


def save(name, rec, profiles, split=None, note="", xs_dir=None):
    tbytes, cbytes, dumps = run_reference(rec, profiles, split, xs_dir)
    out = dict(tape3=np.frombuffer(tbytes, np.uint8), case=np.frombuffer(cbytes, np.uint8),
               nprof=np.int32(len(dumps)), note=np.array(note))
    if xs_dir:   # the synthetic cross-section library travels with the fixture (data files, written by monortm_amd/xsec.py)
        for f in sorted(os.listdir(xs_dir)):
            out["xsfile_" + f] = np.frombuffer(open(os.path.join(xs_dir, f), "rb").read(), np.uint8)
    for i, dmp in enumerate(dumps):
        for k in ("o", "o_by_mol", "oc", "o_clw", "rup", "rdn", "trtot", "rad", "tb", "tmr"):
            out[f"p{i}_{k}"] = getattr(dmp, k)
        out[f"p{i}_tmpsfc_out"] = np.float64(dmp.tmpsfc_out)
        if dmp.odxsec is not None:
            out[f"p{i}_odxsec"] = dmp.odxsec
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    tb = dumps[0].tb
    print(f"{name:28s} {os.path.getsize(path)/1024:8.1f} KiB  nprof={len(dumps)} TB[{tb.min():.2f},{tb.max():.2f}]")


def gen_c2():
    save("c2_base", synth.synthetic_lines(500), [synth.c2_profile()],
         note="BASELINE config 2: 1 profile x 64 layers x 50 channels x 500 lines, downwelling")


def gen_c2_lc_sdep():
    rec = synth.synthetic_lines(300, seed=7, sdep_frac=0.3, lc_frac=0.6)
    a = synth.standard_atmosphere(24)
    wn = synth.c2_channels(16, seed=3)
    pr = synth.Profile(wn=wn, p=a["p"], t=a["t"], tz=a["tz"], wkl=a["wkl"], wbrodl=a["wbrodl"], clw=a["clw"], irt=3)
    save("lc_o2_random", rec, [pr], note="random list, 60% of O2 lines first-order line coupled (IFLG=1/-1), 30% sdep")


def voigt_rows():
    rng = np.random.default_rng(11)
    rows = []
    specs = [(1, 0.7417, 3e-25), (1, 6.1146, 2e-24), (3, 3.671, 4e-22), (7, 1.9835, 4e-27), (7, 3.961, 3e-27),
             (2, 5.22, 8e-25), (4, 2.513, 5e-22), (3, 12.33, 6e-22), (1, 18.577, 5e-24), (7, 14.17, 2e-27),
             (1, 25.085, 2e-24), (3, 30.2, 3e-22)]
    for i, (mol, v, s) in enumerate(specs):
        rows.append(dict(vnu=v, s=s, alfa=rng.uniform(0.04, 0.1), hwhm=rng.uniform(0.05, 0.45) if mol != 7 else 0.05,
                         epp=rng.uniform(0, 1500), n=rng.uniform(0.45, 0.78), shift=rng.uniform(-0.002, 0.002), mol=mol,
                         sdep=(0.0 if i % 3 else rng.uniform(0.06, 0.14))))
    rows.sort(key=lambda r: r["vnu"])
    return rows


def gen_voigt():
    rows = voigt_rows()
    rec = rec_from(rows)
    a = deep_atmosphere(16)
    wn = []
    ks = [0.0, 0.21, 0.9, 2.3, 5.7, 13.0, 37.0, 88.0, 140.0]
    for r in rows[::2]:
        ad = alpha_d(r["vnu"], 230.0, r["mol"])
        for k in ks:
            wn.append(r["vnu"] + k * ad)
            wn.append(r["vnu"] - 0.63 * k * ad)
    wn = np.unique(np.array(wn))
    pr = synth.Profile(wn=wn, p=a["p"], t=a["t"], tz=a["tz"], wkl=a["wkl"], wbrodl=a["wbrodl"], clw=a["clw"], irt=3)
    save("voigt_regions", rec, [pr],
         note="16 layers 1013 -> 0.004 mbar, channels at 0..140 Doppler widths from line centres: Voigt + SD-Voigt, "
              "all Humlicek regions, Lorentz/Voigt switch")


def lc_rows():
    rng = np.random.default_rng(5)

    def yg(sy, sg):
        y = sy * np.array([1.35, 1.15, 1.0, 0.88])
        g = sg * np.array([1.6, 1.25, 1.0, 0.8])
        return (y, g)

    rows = []
    # O2: zero-frequency (non-resonant) band with IFLG=3, resonant lines with IFLG=1
    rows.append(dict(vnu=1.0e-6 + 0.0002, s=4e-33, alfa=0.05, hwhm=0.05, epp=2.1, n=0.8, shift=0.0, mol=7, iflg=3,
                     lc=[yg(0.05, 0.0)]))
    for v in (1.6, 1.87, 1.9, 1.95, 2.0, 2.05, 3.96, 14.2, 23.9, 41.0):
        rows.append(dict(vnu=v, s=10 ** rng.uniform(-27.5, -26.3), alfa=rng.uniform(0.045, 0.055), hwhm=rng.uniform(0.045, 0.055),
                         epp=rng.uniform(0, 900), n=0.8, shift=0.0, mol=7, iflg=1,
                         lc=[yg(rng.uniform(-0.5, 0.5), rng.uniform(-0.03, 0.03))]))
    rows.append(dict(vnu=2.2, s=3e-27, alfa=0.05, hwhm=0.05, epp=300.0, n=0.8, shift=0.0, mol=7))
    rows.append(dict(vnu=47.0, s=3e-27, alfa=0.05, hwhm=0.05, epp=300.0, n=0.8, shift=0.0, mol=7))  # beyond 25 cm-1 of most channels
    # CO2 with and without coupling
    rows.append(dict(vnu=9.3, s=5e-25, alfa=0.07, hwhm=0.09, epp=200.0, n=0.7, shift=-0.001, mol=2, iflg=1,
                     lc=[yg(0.02, 0.001)]))
    rows.append(dict(vnu=9.9, s=5e-25, alfa=0.07, hwhm=0.09, epp=250.0, n=0.7, shift=-0.001, mol=2, iflg=3,
                     lc=[yg(0.01, 0.002)]))
    rows.append(dict(vnu=11.1, s=4e-25, alfa=0.07, hwhm=0.09, epp=120.0, n=0.72, shift=0.001, mol=2))
    # generic molecules with coupling records (O3, H2O)
    rows.append(dict(vnu=7.4, s=4e-22, alfa=0.08, hwhm=0.1, epp=90.0, n=0.7, shift=0.001, mol=3, iflg=1,
                     lc=[yg(-0.03, 0.004)]))
    rows.append(dict(vnu=0.9, s=4e-25, alfa=0.09, hwhm=0.45, epp=400.0, n=0.65, shift=-0.002, mol=1, iflg=3,
                     lc=[yg(0.04, -0.003)]))
    rows.append(dict(vnu=12.7, s=8e-25, alfa=0.095, hwhm=0.0, epp=450.0, n=0.6, shift=0.002, mol=1))  # self width 0 -> 5*alfa
    rows.append(dict(vnu=30.6, s=6e-25, alfa=0.1, hwhm=0.4, epp=130.0, n=0.7, shift=0.0, mol=1))
    rows.sort(key=lambda r: r["vnu"])
    return rows


def gen_lc():
    rows = lc_rows()
    rec = rec_from(rows)
    a = deep_atmosphere(14, ptop=0.02)
    base = np.array([0.4, 0.9, 1.59, 1.88, 1.93, 1.999, 2.04, 3.0, 3.957, 7.41, 9.31, 9.95, 11.0, 12.7, 14.3, 18.0, 22.2,
                     23.91, 26.0, 30.6, 33.0])
    ad = alpha_d(1.9, 230., 7)
    wn = np.unique(np.concatenate([base, 1.9 + ad * np.array([0.4, 3.0, 30.0]), 9.3 + alpha_d(9.3, 230., 2) * np.array([0.5, 8.0]),
                                   7.4 + alpha_d(7.4, 230., 3) * np.array([0.3, 4.0]), 0.9 + alpha_d(0.9, 230., 1) * np.array([0.7, 20.])]))
    prs = [synth.Profile(wn=wn, p=a["p"], t=a["t"], tz=a["tz"], wkl=a["wkl"], wbrodl=a["wbrodl"], clw=a["clw"], irt=3),
           synth.Profile(wn=wn, p=a["p"], t=a["t"] + 4.0, tz=a["tz"] + 4.0, wkl=a["wkl"], wbrodl=a["wbrodl"], clw=a["clw"], irt=3,
                         sclcpl=0.87, sclhw=1.1, y0res=0.002)]
    save("line_coupling", rec, prs,
         note="O2 IFLG=1/3 (-1/-3 records), CO2 and generic molecules with coupling, H2O with self width 0; Lorentz and Voigt; "
              "2nd profile with SCLCPL/SCLHW/Y0RES != defaults")


def gen_cloud_up():
    rec = synth.synthetic_lines(200, seed=21, vhi=40.0)
    wn = np.sort(np.random.default_rng(4).uniform(0.3, 6.5, 24))
    prs = [synth.perturbed_profile(i, wn, nlay=32, cloud=True, irt=irt) for i, irt in enumerate((3, 1, 1, 2))]
    save("cloud_updown", rec, prs, note="4 profiles x 32 layers x 24 channels with TKC liquid cloud; IRT=3,1,1,2 "
                                        "(upwelling with TBOUND 290, emis .6, refl .4; limb)")


def gen_grid_ir():
    """DVSET != 0 grid, NMOL=22, wavenumbers up to ~1000 cm-1: foreign-continuum
    formula branch (>600 cm-1), Rayleigh (>=820), TAPE3 block skip/stop logic."""
    rng = np.random.default_rng(33)
    rows = []
    for v in np.sort(rng.uniform(840.0, 1010.0, 700)):
        mol = int(rng.choice([1, 1, 2, 3, 3, 4, 7, 7]))  # molecules > 7 hit an out-of-bounds read in the reference (modm.f90:845)
        rows.append(dict(vnu=float(v), s=10 ** rng.uniform(-26, -22.5) * (1e-3 if mol in (7, 22) else 1.0) * (1e3 if mol in (3, 4) else 1),
                         alfa=rng.uniform(0.04, 0.1), hwhm=rng.uniform(0.05, 0.4) if mol not in (7, 22) else 0.045,
                         epp=rng.uniform(0, 1800), n=rng.uniform(0.5, 0.78), shift=rng.uniform(-0.004, 0.001), mol=mol,
                         iso=int(rng.choice([1, 1, 1, 2])) if mol in (1, 2, 3) else 1))
    rec = rec_from(rows)
    a = deep_atmosphere(12, ptop=1.0, nmol=22)
    v1, dv, n = 900.0, 0.05, 160
    wn = v1 + dv * np.arange(n)
    pr = synth.Profile(wn=wn, p=a["p"], t=a["t"], tz=a["tz"], wkl=a["wkl"], wbrodl=a["wbrodl"], clw=a["clw"], irt=1,
                       dvset=dv, tmpsfc=294.0, emiss=np.full(n, 0.98), reflc=np.full(n, 0.02))
    save("ir_grid_nmol22", rec, [pr], split=[100, 230, 600],
         note="900-908 cm-1 DVSET=0.05 grid, NMOL=22, isotopologues 1-2, 5 TAPE3 blocks (first ones skipped by v1-25)")


def gen_cntnm_factors():
    rec = synth.synthetic_lines(60, seed=99)
    a = synth.standard_atmosphere(10, ztop_km=20)
    wn = np.array([0.8, 3.0, 6.1, 9.9, 15.0, 22.235, 29.0, 36.5, 44.0, 53.7])
    prs = []
    for fac in ([1, 1, 1, 1, 1, 1, 1], [0, 1, 1, 1, 1, 1, 1], [1, 0, 1, 1, 1, 1, 1], [0, 0, 0, 0, 0, 0, 0],
                [0.5, 1.7, 2.0, 1, 1, 0.3, 1]):
        prs.append(synth.Profile(wn=wn, p=a["p"], t=a["t"], tz=a["tz"], wkl=a["wkl"], wbrodl=a["wbrodl"], clw=a["clw"],
                                 irt=3, cntnm=np.array(fac, float)))
    save("cntnm_factors", rec, prs, note="continuum scale factors (ICNTNM variants and XSELF..XRAYL scaling)")


def gen_cut_boundaries():
    """Exact equality at every threshold of the Lorentz path: |WN - Xnu| = 25 (live, the test is "> 25", modm.f90:384),
    WN + Xnu = 25 (negative resonance included, "<= 0", modm.f90:713), WN = Xnu (line centre), for generic molecules,
    uncoupled O2 and CO2.  Zero pressure shift, so the shifted centres stay exactly representable."""
    def row(v, mol, s):
        return dict(vnu=float(v), s=s, alfa=0.08, hwhm=0.3, epp=300.0, n=0.7, shift=0.0, mol=mol)
    rows = [row(5.0, 1, 3e-24), row(12.5, 1, 2e-24), row(15.0, 1, 1e-24), row(30.0, 1, 4e-24), row(35.0, 1, 5e-24),
            row(45.0, 1, 2e-24),
            row(55.0, 2, 1e-25), row(37.5, 2, 1e-25),
            row(10.0, 3, 3e-22), row(15.0, 3, 2e-22), row(37.5, 3, 1e-22),
            row(5.0, 7, 1e-28), row(20.0, 7, 2e-28), row(45.0, 7, 1e-28)]
    rows.sort(key=lambda r: r["vnu"])
    rec = rec_from(rows)
    a = synth.standard_atmosphere(6, ztop_km=25)
    wn = np.array([5.0, 10.0, 12.5, 20.0, 30.0])
    prs = [synth.Profile(wn=wn, p=a["p"], t=a["t"], tz=a["tz"], wkl=a["wkl"], wbrodl=a["wbrodl"], clw=a["clw"], irt=3)]
    # the same lines seen from a DVSET grid whose points hit the thresholds too (grid-mode continuum)
    wn2 = 5.0 + 2.5 * np.arange(11)
    prs.append(synth.Profile(wn=wn2, p=a["p"], t=a["t"], tz=a["tz"], wkl=a["wkl"], wbrodl=a["wbrodl"], clw=a["clw"], irt=3,
                             dvset=2.5))
    save("cut_boundaries", rec, prs[:1], note="channels exactly 25 cm-1 from line centres, WN+Xnu exactly 25, WN exactly on a centre")
    save("cut_boundaries_grid", rec, prs[1:], note="the same thresholds hit by a 2.5 cm-1 DVSET grid")


def gen_temperature_brackets():
    """Layer temperatures exactly on the brackets of the line-coupling table (250, 296 K: ILC selection, modm.f90:301-309),
    at 200 / 340 K (its ends), near the limits of TIPS (72 K, 2950 K), and layers with zero column of some molecules
    (W_SPECIES = 0 shortcut, modm.f90:318-321); coupled O2 / CO2 / generic lines."""
    rows = lc_rows()
    rec = rec_from(rows)
    temps = np.array([296.0, 250.0, 200.0, 340.0, 72.0, 2950.0, 249.999, 296.001, 273.15, 1000.0])
    nlay = len(temps)
    a = synth.standard_atmosphere(nlay, ztop_km=20)
    wkl = a["wkl"].copy()
    wkl[2, 0] = 0.0     # no water vapour in layer 3
    wkl[5, 6] = 0.0     # no O2 in layer 6
    wkl[7, 1:4] = 0.0   # no CO2 / O3 / N2O in layer 8
    tz = np.concatenate([[temps[0]], 0.5 * (temps[:-1] + temps[1:]), [temps[-1]]])
    wn = np.array([0.9, 1.9, 1.999, 3.957, 7.41, 9.31, 12.7, 22.2, 30.6])
    prs = [synth.Profile(wn=wn, p=a["p"], t=temps, tz=tz, wkl=wkl, wbrodl=a["wbrodl"], clw=a["clw"], irt=3),
           synth.Profile(wn=wn, p=a["p"], t=temps, tz=tz, wkl=wkl, wbrodl=a["wbrodl"], clw=a["clw"], irt=1, tmpsfc=300.0,
                         emiss=np.full(len(wn), 0.95), reflc=np.full(len(wn), 0.05))]
    save("temperature_brackets", rec, prs,
         note="layer temperatures on the coupling-table brackets and at the TIPS limits; zero columns in single layers")


def gen_ibrd():
    rng = np.random.default_rng(77)
    rows = []
    for v in np.sort(rng.uniform(1.0, 30.0, 40)):
        mol = int(rng.choice([1, 2, 3, 4, 7]))
        flg = [int(x) for x in (rng.random(7) < 0.4)]
        dat = []
        for j in range(7):
            dat += [rng.uniform(0.03, 0.15), rng.uniform(0.4, 0.8), rng.uniform(-0.004, 0.004)]
        rows.append(dict(vnu=float(v), s=10 ** rng.uniform(-26, -24) * (1e-3 if mol == 7 else 1) * (1e3 if mol in (3, 4) else 1),
                         alfa=rng.uniform(0.04, 0.1), hwhm=rng.uniform(0.05, 0.4) if mol != 7 else 0.05, epp=rng.uniform(0, 1500),
                         n=rng.uniform(0.5, 0.78), shift=rng.uniform(-0.003, 0.003), mol=mol, brd_flg=flg, brd_dat=dat))
    rec = rec_from(rows)
    a = synth.standard_atmosphere(12, ztop_km=30)
    wn = np.sort(rng.uniform(0.5, 30.0, 14))
    prs = [synth.Profile(wn=wn, p=a["p"], t=a["t"], tz=a["tz"], wkl=a["wkl"], wbrodl=a["wbrodl"], clw=a["clw"], irt=3, ibrd=ib)
           for ib in (1, 0)]
    # two separate reference processes are not needed: IBRD only changes per-call arithmetic
    save("ibrd_species_broadening", rec, prs, note="IBRD=1 species-by-species broadening/shift data (then IBRD=0 on the same file)")


def _ir_case(name, vlo, vhi, nwn, seed, irt=3, grid=False, note=""):
    """One spectral window above the microwave: a handful of lines inside it + every continuum branch that
    acts there (src/contnm.f90:536-1068)."""
    rng = np.random.default_rng(seed)
    rows = []
    for v in np.sort(rng.uniform(vlo + 1.0, vhi - 1.0, 40)):
        mol = int(rng.choice([1, 2, 3, 4, 7]))
        rows.append(dict(vnu=float(v), s=10 ** rng.uniform(-26, -23) * (1e-3 if mol == 7 else 1) * (1e2 if mol in (3, 4) else 1),
                         alfa=rng.uniform(0.04, 0.1), hwhm=rng.uniform(0.05, 0.4) if mol != 7 else 0.045,
                         epp=rng.uniform(0, 1500), n=rng.uniform(0.5, 0.78), shift=rng.uniform(-0.004, 0.001), mol=mol))
    rec = rec_from(rows)
    a = deep_atmosphere(8, ptop=5.0)
    if grid:
        dv = (vhi - vlo) / (nwn - 1)
        wn = vlo + dv * np.arange(nwn)
    else:
        dv = 0.0
        wn = np.sort(rng.uniform(vlo, vhi, nwn))
    kw = dict(tmpsfc=288.0, emiss=np.full(nwn, 0.97), reflc=np.full(nwn, 0.03)) if irt == 1 else {}
    pr = synth.Profile(wn=wn, p=a["p"], t=a["t"], tz=a["tz"], wkl=a["wkl"], wbrodl=a["wbrodl"], clw=a["clw"], irt=irt,
                       dvset=dv, **kw)
    save(name, rec, [pr], note=note)


def gen_ir_uv():
    _ir_case("ir_o2_fundamental", 1330.0, 1870.0, 36, 101, irt=1, note="O2 collision-induced fundamental 1340-1850 cm-1")
    _ir_case("ir_n2_fundamental_co2", 1990.0, 3010.0, 48, 102, irt=1,
             note="N2 fundamental 2001-2898, CO2 continuum scaling 2000-2998 and band-head T dependence 2386-2434")
    _ir_case("ir_n2_overtone", 4300.0, 4950.0, 24, 103, note="N2 first overtone 4340-4910")
    _ir_case("nir_o2_bands", 7500.0, 11050.0, 60, 104, note="O2 1.27 micron (7536-8500), O2 1.06 micron analytic (9100-11000), O3 Chappuis tail")
    _ir_case("vis_o2_aband_chappuis", 12900.0, 16800.0, 48, 105, irt=1, note="O2 A band continuum, O2 visible, O3 Chappuis/Wulf")
    _ir_case("vis_grid_chappuis", 17000.0, 17400.0, 81, 106, grid=True, note="DVSET grid inside the Chappuis band + O2 visible")
    _ir_case("uv_hartley_huggins", 27000.0, 31500.0, 40, 107, note="O3 Hartley-Huggins with T dependence, end of O2 visible (29870)")
    _ir_case("uv_40800_seam", 36500.0, 41500.0, 48, 108, note="O2 Herzberg from 36000, O3 HH/UV seam at 40800 (I_FIX logic)")
    _ir_case("fuv_schumann_runge", 53000.0, 57800.0, 36, 109, note="O3 UV up to 54000, O2 Herzberg, O2 far-UV from 56740")


def gen_sgl_cloud():
    """Single-precision reference build (harness_ref_sgl, the reference's "sgl" flag set) on the cloud / up-down
    batch: the fixture for a REAL*4 caller of the drop-in modules (BASELINE config 5 flavour)."""
    global HARNESS
    keep = HARNESS
    HARNESS = os.path.join(ROOT, "oracle", "_ref", "harness_ref_sgl")
    try:
        rec = synth.synthetic_lines(200, seed=21, vhi=40.0)
        wn = np.sort(np.random.default_rng(4).uniform(0.3, 6.5, 24))
        prs = [synth.perturbed_profile(i, wn, nlay=32, cloud=True, irt=irt) for i, irt in enumerate((3, 1, 1, 2))]
        save("sgl_cloud_updown", rec, prs, note="same inputs as cloud_updown, outputs of the SINGLE-PRECISION reference build")
    finally:
        HARNESS = keep


def gen_sgl_more():
    """The inputs of five double-precision fixtures replayed through the SINGLE-PRECISION reference build (a float
    prepare stage of the real_kind = 4 kernels must be judged against the sgl build, not only "5e-5 of dbl").  Outputs are the
    REAL*4 values of the reference, stored as float32 (exact)."""
    sgl = os.path.join(ROOT, "oracle", "_ref", "harness_ref_sgl")
    for name in ("c2_base", "line_coupling", "voigt_regions", "lc_o2_random", "all_molecules"):
        z = np.load(os.path.join(HERE, name + ".npz"))
        with tempfile.TemporaryDirectory() as d:
            tp, cp, op = (os.path.join(d, n) for n in ("TAPE3", "case.bin", "out.bin"))
            open(tp, "wb").write(z["tape3"].tobytes())
            open(cp, "wb").write(z["case"].tobytes())
            r = subprocess.run([sgl, cp, tp, op], cwd=d, capture_output=True, text=True)
            if r.returncode != 0 or "HARNESS_SECONDS" not in r.stdout:
                # voigt_regions: the REAL*4 build of SDVOIGT returns Re(v) < 0 for a speed-dependent line of the thin upper
                # layers and the reference STOPs (src/modm.f90:1062) - that case has no single-precision answer
                print(f"sgl_{name}: the single-precision reference stops on these inputs ({(r.stdout + r.stderr).strip()[-60:]!r}); no fixture")
                continue
            dumps = caseio.read_dump(op)
        out = dict(tape3=z["tape3"], case=z["case"], nprof=np.int32(len(dumps)),
                   note=np.array(f"inputs of {name}, outputs of the SINGLE-PRECISION reference build (harness_ref_sgl)"))
        worst = 0.0
        for i, dmp in enumerate(dumps):
            for k in ("o", "o_by_mol", "oc", "o_clw", "rup", "rdn", "trtot", "rad", "tb", "tmr"):
                v = getattr(dmp, k)
                assert np.array_equal(v.astype(np.float32).astype(np.float64), v, equal_nan=True), (name, k)
                out[f"p{i}_{k}"] = v.astype(np.float32)
            out[f"p{i}_tmpsfc_out"] = np.float64(dmp.tmpsfc_out)
            d64 = z[f"p{i}_tb"]
            ok = np.isfinite(d64) & np.isfinite(dmp.tb)
            worst = max(worst, float(np.max(np.abs(dmp.tb[ok] - d64[ok]) / np.abs(d64[ok]))) if ok.any() else 0.0)
        path = os.path.join(HERE, "sgl_" + name + ".npz")
        np.savez_compressed(path, **out)
        print(f"sgl_{name:24s} {os.path.getsize(path)/1024:8.1f} KiB  nprof={len(dumps)}  TB sgl vs dbl: {worst:.2e}  "
              f"NaN in TB: {int(sum(np.isnan(dm.tb).sum() for dm in dumps))}")


def gen_self_coupling():
    """IFLG = 5 lines (XG = -5): a foreign and a self coupling record follow the line.  The reference recognises the self
    record only when the PREVIOUS record is a -5 one too (src/modm.f90:339), so the first -5 line of a run is treated as
    foreign-only and its self record is then walked as if it were a line - reproduced literally.
    An uncoupled line comes first in each molecule, so that XG(I,J-1) is a defined read.  The G values of the self sets are
    sized like line strengths because the walked self record uses G(200 K) as its S0."""
    rng = np.random.default_rng(55)

    def yg(sy, sg):
        y = sy * np.array([1.3, 1.12, 1.0, 0.9])
        # the walked self record takes its isotopologue from the MOL word, which holds the bits of Y(296 K) as REAL*4
        # (src/lnfl_mod.f90:67,80-82): nudge Y(296 K) by < 1e-4 relative so that those bits give isotopologue 1 - otherwise
        # the reference indexes scor(i,0), an out-of-bounds read that no fixture could pin
        b = int(np.float32(y[2]).view(np.int32))
        b += (150 - b % 1000) % 1000
        y[2] = float(np.int32(b).view(np.float32))
        assert (int(np.float32(y[2]).view(np.int32)) % 1000) // 100 == 1
        return (y, sg * np.array([1.5, 1.2, 1.0, 0.85]))

    rows = [dict(vnu=5.1, s=3e-25, alfa=0.07, hwhm=0.09, epp=150.0, n=0.7, shift=0.0, mol=2),
            dict(vnu=2.9, s=2e-22, alfa=0.08, hwhm=0.1, epp=60.0, n=0.7, shift=0.001, mol=3)]
    for v in (8.2, 8.9, 9.6, 10.4, 15.0):     # CO2 Q-branch-like run with foreign + self sets
        rows.append(dict(vnu=v, s=10 ** rng.uniform(-24.8, -24.2), alfa=0.07, hwhm=0.09, epp=rng.uniform(50, 500), n=0.72,
                         shift=-0.001, mol=2, iflg=5, lc=[yg(rng.uniform(0.01, 0.03), 0.001), yg(rng.uniform(0.02, 0.05), 2e-27)]))
    for v in (6.3, 6.8, 7.7):                 # a generic molecule with -5 sets
        rows.append(dict(vnu=v, s=10 ** rng.uniform(-21.8, -21.2), alfa=0.08, hwhm=0.1, epp=rng.uniform(50, 300), n=0.7,
                         shift=0.001, mol=3, iflg=5, lc=[yg(rng.uniform(-0.04, 0.04), 0.003), yg(rng.uniform(0.02, 0.06), 4e-24)]))
    rows += [dict(vnu=20.0, s=5e-25, alfa=0.09, hwhm=0.4, epp=200.0, n=0.65, shift=0.0, mol=1),
             dict(vnu=12.0, s=3e-27, alfa=0.05, hwhm=0.05, epp=300.0, n=0.8, shift=0.0, mol=7)]
    rows.sort(key=lambda r: r["vnu"])
    rec = rec_from(rows)
    a = deep_atmosphere(12, ptop=0.05)
    wn = np.array([0.5, 2.9, 5.1, 6.3, 6.79, 7.7, 8.21, 8.9, 9.3, 9.61, 10.4, 12.0, 15.0, 19.0, 24.0, 31.0])
    prs = [synth.Profile(wn=wn, p=a["p"], t=a["t"], tz=a["tz"], wkl=a["wkl"], wbrodl=a["wbrodl"], clw=a["clw"], irt=3),
           synth.Profile(wn=wn, p=a["p"], t=a["t"] - 6.0, tz=a["tz"] - 6.0, wkl=a["wkl"], wbrodl=a["wbrodl"], clw=a["clw"], irt=1,
                         tmpsfc=285.0, emiss=np.full(len(wn), 0.9), reflc=np.full(len(wn), 0.1), sclcpl=0.9)]
    save("self_coupling_m5", rec, prs, note="IFLG=5 / XG=-5 lines of CO2 and O3 with foreign and self coupling records, incl. the "
                                          "reference's first-of-run mis-walk (Appendix E.2)")


sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import ALLMOL_SKIP, TIPS_ISONM  # noqa: E402


def all_molecule_vmr(nlay, zmid, nmol=39):
    """Mixing ratios for all 39 HITRAN molecules: the seven majors as in deep_atmosphere(), N2 as molecule 22, trace species
    1e-9 .. 1e-7 with a smooth height dependence (so that no two columns are proportional)."""
    vmr = np.zeros((nlay, nmol))
    vmr[:, 0] = 0.008 * np.exp(-zmid / 2.0) + 5e-6
    vmr[:, 1] = 4e-4
    vmr[:, 2] = 3e-7 * (1 + zmid / 5.0) * np.exp(-np.maximum(zmid - 35, 0) / 8)
    vmr[:, 3] = 3.2e-7
    vmr[:, 4] = 1.5e-7
    vmr[:, 5] = 1.7e-6
    vmr[:, 6] = 0.209
    for m in range(8, nmol + 1):
        vmr[:, m - 1] = 10 ** (-9.0 + 2.0 * ((m * 7) % 11) / 10.0) * (1 + 0.3 * np.sin(zmid / (3.0 + m % 5)))
    vmr[:, 21] = 0.781
    return vmr


def all_molecule_rows(rng, vmr_col, skip=(), vlo=1.0, vhi=50.0):
    """One line per (molecule, isotopologue <= min(9, ISONM)) - every slot TIPS_2003 fills.  Molecules > 7 get hwhm = alfa: the
    reference indexes its 7-element rho_molec with the molecule number (modm.f90:845), and with equal self and foreign widths
    that out-of-bounds value multiplies zero."""
    slots = [(m, i) for m in range(1, 40) if m not in skip for i in range(1, min(9, TIPS_ISONM[m - 1]) + 1)]
    centres = np.linspace(vlo, vhi, len(slots)) + rng.uniform(-0.1, 0.1, len(slots))
    order = rng.permutation(len(slots))
    rows = []
    for v, k in zip(centres, order):
        m, i = slots[k]
        alfa = rng.uniform(0.05, 0.1)
        hwhm = rng.uniform(0.06, 0.4) if m <= 6 else (0.05 if m == 7 else alfa)
        if m == 7:
            alfa = 0.05
        # peak layer optical depth ~ 0.01-0.1 for every species whatever its abundance
        s = 0.02 * np.pi * 0.08 / max(vmr_col[m - 1], 1e-30) * 10 ** rng.uniform(-0.5, 0.5)
        rows.append(dict(vnu=float(v), s=float(s), alfa=alfa, hwhm=hwhm, epp=rng.uniform(0, 1200), n=rng.uniform(0.5, 0.78),
                         shift=rng.uniform(-0.002, 0.002), mol=m, iso=i, sdep=(rng.uniform(0.05, 0.12) if k % 7 == 0 else 0.0)))
    rows.sort(key=lambda r: r["vnu"])
    return rows


# (ALLMOL_SKIP: molecules whose HALFWHM_C picks up a NaN from beyond rho_molec(7) in the -O0 flang build of the reference even
# with hwhm = alfa (NaN * 0): they cannot be pinned and stay out of the fixture - DESIGN.md section 4, deviation 1)


def gen_all_molecules():
    rng = np.random.default_rng(3939)
    nlay = 14
    a = deep_atmosphere(nlay, ptop=0.05)
    plev = np.exp(np.linspace(np.log(1013.0), np.log(0.05), nlay + 1))
    zmid = -7.0 * np.log(np.sqrt(plev[:-1] * plev[1:]) / 1013.0)
    air = 2.1e25 * (plev[:-1] - plev[1:]) / 1013.0
    vmr = all_molecule_vmr(nlay, zmid)
    wkl = vmr * air[:, None]
    wbrodl = 0.0093 * air
    rows = all_molecule_rows(rng, wkl[0], skip=ALLMOL_SKIP)
    rec = rec_from(rows)
    cent = np.array([r["vnu"] for r in rows])
    # a channel near every centre (Lorentz core in the troposphere, Voigt aloft) and some between the lines
    wn = np.unique(np.concatenate([cent[::2] + 0.01, cent[1::2] - 2e-5, np.linspace(0.7, 52.0, 12)]))
    # First profile: columns of molecules 8-39 zero (W_SPECIES = 0 shortcut, modm.f90:318-321, for 32 molecules).  It also
    # takes the reference's very first LINES call: on that call the stack slot behind rho_molec(7) still holds start-up
    # garbage that reads as NaN (observed: molecule 8, first layer, first wavenumber only); afterwards the slot holds a stale
    # finite local of the previous call, which hwhm = alfa multiplies by zero.
    wkl0 = wkl.copy()
    wkl0[:, 7:] = 0.0
    wkl0[:, 21] = wkl[:, 21]
    prs = [synth.Profile(wn=wn[::5], p=a["p"], t=a["t"] - 3.0, tz=a["tz"] - 3.0, wkl=wkl0, wbrodl=wbrodl, clw=a["clw"], irt=3),
           synth.Profile(wn=wn, p=a["p"], t=a["t"], tz=a["tz"], wkl=wkl, wbrodl=wbrodl, clw=a["clw"], irt=3),
           synth.Profile(wn=wn[::3], p=a["p"], t=a["t"] + 7.0, tz=a["tz"] + 7.0, wkl=wkl * 1.1, wbrodl=wbrodl, clw=a["clw"], irt=1,
                         tmpsfc=291.0, emiss=np.full(len(wn[::3]), 0.93), reflc=np.full(len(wn[::3]), 0.07))]
    save("all_molecules", rec, prs, note="NMOL = 39: one line per (molecule, isotopologue <= 9) TIPS_2003 knows, hwhm = alfa for molecules > 7 "
                                         "(modm.f90:845 out-of-bounds term vanishes), trace columns; 3 profiles (21 channels with majors only, 102 and 34 channels with all)")


def _xsec_case(name, nlay, ptop, wn, irt, seed, note, tshift=0.0, fscdxs_pad=(0.0, 0.0)):
    """IXSECT = 1: cross-section molecules CCL4, F11, F12 from a synthetic FSCDXS / xs library (monortm_amd/xsec.py) on top of a
    few lines and the infrared continuum.  Every layer lies above the pressures of the measurements (the reference's
    convolve() overruns its 10^7-element work array otherwise, see xsec.synthetic_library); the TOP layer sits 5 % above the
    pressure of the F12 measurement, so that it takes the linearly interpolated values (extra Lorentz width below a tenth of
    the measurement's, src/monortm_sub.F90:1787,:1822-1828) while the lower ones are convolved (:1788-1821).  Layer
    temperatures below, between and above the tabulated ones.  ONE profile per reference process: a second
    MONORTM_XSEC_SUB call re-opens units that are still connected (:1662)."""
    from monortm_amd import xsec

    rng = np.random.default_rng(seed)
    rows = []
    for v in np.sort(rng.uniform(wn[0] - 5, wn[-1] + 5, 30)):
        mol = int(rng.choice([1, 2, 3, 4]))
        rows.append(dict(vnu=float(v), s=10 ** rng.uniform(-26, -23.5) * (1e2 if mol in (3, 4) else 1), alfa=rng.uniform(0.04, 0.1),
                         hwhm=rng.uniform(0.05, 0.4), epp=rng.uniform(0, 1500), n=rng.uniform(0.5, 0.78), shift=rng.uniform(-0.004, 0.001), mol=mol))
    rec = rec_from(rows)
    a = deep_atmosphere(nlay, ptop=ptop)
    with tempfile.TemporaryDirectory() as xd:
        names = xsec.synthetic_library(xd, f12_pres_mb=float(a["p"][-1]) / 1.05, fscdxs_pad=fscdxs_pad)
        air = a["wbrodl"] / 0.781
        xamnt = np.stack([air * 1.0e-10 * (1 + 0.2 * np.cos(np.arange(nlay))), air * 2.6e-10, air * 5.3e-10 * np.exp(-np.arange(nlay) / 9.0)], axis=1)
        kw = dict(tmpsfc=289.0, emiss=np.full(len(wn), 0.98), reflc=np.full(len(wn), 0.02)) if irt == 1 else {}
        pr = synth.Profile(wn=wn, p=a["p"], t=a["t"] + tshift, tz=a["tz"] + tshift, wkl=a["wkl"], wbrodl=a["wbrodl"], clw=a["clw"], irt=irt,
                           xs_names=names, xamnt=xamnt, **kw)
        save(name, rec, [pr], note=note, xs_dir=xd)


def gen_xsec():
    rng = np.random.default_rng(41)
    wn1 = np.sort(np.concatenate([rng.uniform(772.0, 811.0, 14), rng.uniform(831.0, 859.0, 10), rng.uniform(860.0, 948.0, 8), [790.0, 846.0]]))
    _xsec_case("xsec_ccl4_f11_f12", 10, 115.0, wn1, 1, 5,
               "IXSECT=1: CCL4 (3 temperatures), F11 (2 regions; one file in mbar), F12 (1 temperature); 770-950 cm-1, 10 layers to 115 mbar")
    wn2 = np.sort(np.concatenate([rng.uniform(1061.0, 1106.0, 12), rng.uniform(900.0, 949.0, 6), [1085.0, 921.0, 1200.0, 700.0]]))
    _xsec_case("xsec_two_regions_down", 7, 200.0, wn2, 3, 6,
               "IXSECT=1: F11's second region (1060-1107), F12, channels outside every region and outside the regions' 1 cm-1 "
               "margins; downwelling, 7 layers to 200 mbar, temperatures +12 K (above the warmest table)", tshift=12.0)
    # FSCDXS bounds wider than the file headers (2.3 cm-1 below, 3.6 above): channels inside the header range, between header
    # and FSCDXS bounds (processed, contribute nothing), and within 1 cm-1 outside the FSCDXS bounds only
    wn3 = np.sort(np.concatenate([rng.uniform(831.0, 859.0, 8), rng.uniform(772.0, 811.0, 6),
                                  [828.2, 829.4, 860.4, 862.9, 864.1, 767.0, 768.9, 813.0, 816.2]]))
    _xsec_case("xsec_fscdxs_bounds", 6, 150.0, wn3, 3, 8,
               "IXSECT=1: FSCDXS bounds 2.3 / 3.6 cm-1 wider than the xs file headers - region test on the FSCDXS pair, grid and "
               "in-range test on the header pair (src/monortm_sub.F90:1645 vs :1663-1666)", fscdxs_pad=(2.3, 3.6))


def gen_sgl_grid():
    """DVSET /= 0 through the SINGLE-PRECISION reference build, on the grid its own driver makes: WN(J) = V1 + (J-1)*DVSET
    with V1 REAL*8 and the product in REAL*4 (src/monortm_sub.F90:287), DVSET a REAL*4 value.  Consecutive differences of that
    grid deviate from DVSET by ~6e-8 J DVSET (a per-step test at 1e-6 refused it from point ~20 on)."""
    global HARNESS
    keep = HARNESS
    HARNESS = os.path.join(ROOT, "oracle", "_ref", "harness_ref_sgl")
    try:
        rec = synth.synthetic_lines(300, seed=61, vhi=45.0)
        for name, v1, dv, n in (("sgl_grid_dv05", 0.5, 0.05, 240), ("sgl_grid_dv01", 3.0, 0.01, 520)):
            dv4 = np.float32(dv)
            wn = v1 + (np.arange(n, dtype=np.float32) * dv4).astype(np.float64)   # REAL*4 product, REAL*8 sum
            assert np.max(np.abs(np.diff(wn) - float(dv4))) > 1e-6 * float(dv4)   # the grid the old check refused
            a = synth.standard_atmosphere(12, ztop_km=30)
            clw = np.zeros(12)
            clw[1] = 0.02
            prs = [synth.Profile(wn=wn, p=a["p"], t=a["t"], tz=a["tz"], wkl=a["wkl"], wbrodl=a["wbrodl"], clw=clw, irt=3, dvset=float(dv4)),
                   synth.Profile(wn=wn, p=a["p"], t=a["t"] + 5.0, tz=a["tz"] + 5.0, wkl=a["wkl"] * 1.2, wbrodl=a["wbrodl"], clw=clw, irt=1,
                                 dvset=float(dv4), tmpsfc=290.0, emiss=np.full(n, 0.6), reflc=np.full(n, 0.4))]
            save(name, rec, prs, note=f"SINGLE-PRECISION reference build on the sgl driver's own DVSET grid: V1={v1}, DVSET=REAL*4({dv}), {n} points "
                                      "(REAL*4 product (J-1)*DVSET, src/monortm_sub.F90:287); down- and upwelling, cloud layer")
    finally:
        HARNESS = keep


def _neg_rows(negative=True):
    rng = np.random.default_rng(404)
    rows = []
    specs = [(1, 0.74), (1, 6.11), (1, 18.58), (1, 25.09), (1, 32.9), (1, 47.1), (3, 1.2), (3, 3.67), (3, 9.9), (3, 12.33), (3, 21.0),
             (3, 30.2), (3, 38.8), (4, 2.51), (4, 14.9), (4, 27.6), (2, 5.22), (2, 16.4), (2, 29.9), (7, 1.98), (7, 3.96), (7, 14.17),
             (7, 24.6), (5, 3.85), (5, 7.69), (6, 10.5), (6, 20.9)]
    for i, (mol, v) in enumerate(specs):
        s = 10 ** rng.uniform(-25.5, -23.5) * (1e-3 if mol == 7 else 1.0) * (1e2 if mol in (3, 4) else 1.0)
        if negative and i % 3 == 1:
            s = -s     # unphysical: the reference adds the negative term (src/modm.f90:432), it must not be clamped away
        rows.append(dict(vnu=v, s=s, alfa=rng.uniform(0.04, 0.1), hwhm=rng.uniform(0.05, 0.45) if mol != 7 else 0.05,
                         epp=rng.uniform(0, 1500), n=rng.uniform(0.45, 0.78), shift=rng.uniform(-0.002, 0.002), mol=mol))
    rows.sort(key=lambda r: r["vnu"])
    return rows


def gen_negative_nan():
    """Unphysical inputs that the reference's arithmetic nevertheless defines: negative line strengths (the
    term S~ x shape is added with its sign, src/modm.f90:432; inside the 25 cm-1 window the bracket a2/den - pedestal is then
    NEGATIVE and outside it positive - a clamp-as-test would get both wrong) and a NaN column amount in one layer (WTOT, the
    number-density ratios and with them every width of that layer are NaN: every molecule's optical depth there is NaN)."""
    global HARNESS
    wn = np.array([0.5, 0.74, 1.9, 3.67, 5.2, 6.1, 9.0, 12.3, 14.2, 16.4, 18.6, 21.0, 23.0, 25.1, 27.0, 30.2, 33.0, 36.0, 41.0, 46.0, 52.0])
    a = deep_atmosphere(8, ptop=1.0)
    up = dict(irt=1, tmpsfc=288.0, emiss=np.full(len(wn), 0.9), reflc=np.full(len(wn), 0.1))
    neg = [synth.Profile(wn=wn, p=a["p"], t=a["t"], tz=a["tz"], wkl=a["wkl"], wbrodl=a["wbrodl"], clw=a["clw"], irt=3),
           synth.Profile(wn=wn, p=a["p"], t=a["t"] + 3.0, tz=a["tz"] + 3.0, wkl=a["wkl"], wbrodl=a["wbrodl"], clw=a["clw"], **up)]
    # the same on a dense DVSET grid (two wavenumbers per lane, far field): 700 points from 2 to 30 cm-1
    dv = 0.04
    wng = 2.0 + dv * np.arange(700)
    neg.append(synth.Profile(wn=wng, p=a["p"][:3], t=a["t"][:3], tz=a["tz"][:4], wkl=a["wkl"][:3], wbrodl=a["wbrodl"][:3], clw=a["clw"][:3],
                             irt=3, dvset=dv))
    save("negative_strength", rec_from(_neg_rows(True)), neg,
         note="every third line with NEGATIVE strength (generic molecules, CO2, O2): terms added with their sign; sparse channels "
              "down / up and a 700-point DVSET grid")
    wkl = a["wkl"].copy()
    wkl[3, 2] = np.nan   # O3 column of layer 4
    nanp = [synth.Profile(wn=wn, p=a["p"], t=a["t"], tz=a["tz"], wkl=wkl, wbrodl=a["wbrodl"], clw=a["clw"], irt=3),
            synth.Profile(wn=wng, p=a["p"][2:5], t=a["t"][2:5], tz=a["tz"][2:6], wkl=wkl[2:5], wbrodl=a["wbrodl"][2:5], clw=a["clw"][2:5],
                          irt=3, dvset=dv)]
    save("nan_column", rec_from(_neg_rows(False)), nanp,
         note="NaN column amount of O3 in one layer: NaN positions and the finite values elsewhere are the reference's "
              "(tests compare NaN patterns exactly; not part of golden_names())")
    keep = HARNESS
    HARNESS = os.path.join(ROOT, "oracle", "_ref", "harness_ref_sgl")
    try:
        save("sgl_negative_strength", rec_from(_neg_rows(True)), neg, note="inputs of negative_strength, outputs of the SINGLE-PRECISION reference build")
        save("nan_sgl_column", rec_from(_neg_rows(False)), nanp, note="inputs of nan_column, outputs of the SINGLE-PRECISION reference build")
    finally:
        HARNESS = keep


def gen_real_like():
    global HARNESS
    rec, kw, info = synth.real_like_file()
    wn = synth.sounder_channels()
    n = len(wn)

    def profiles(ptop):
        a = deep_atmosphere(16, ptop=ptop)
        prs = [synth.Profile(wn=wn, p=a["p"], t=a["t"], tz=a["tz"], wkl=a["wkl"], wbrodl=a["wbrodl"], clw=a["clw"], irt=3),
               synth.Profile(wn=wn, p=a["p"], t=a["t"] + 4.0, tz=a["tz"] + 4.0, wkl=a["wkl"] * 1.15, wbrodl=a["wbrodl"], clw=a["clw"], irt=1,
                             tmpsfc=289.0, emiss=np.full(n, 0.55), reflc=np.full(n, 0.45), ibrd=1)]
        dv = 0.01
        wng = 0.5 + dv * np.arange(1200)
        prs.append(synth.Profile(wn=wng, p=a["p"][[0, 5, 10, 14]], t=a["t"][[0, 5, 10, 14]], tz=np.array([a["tz"][0], a["tz"][5], a["tz"][10], a["tz"][14], a["tz"][15]]),
                                 wkl=a["wkl"][[0, 5, 10, 14]], wbrodl=a["wbrodl"][[0, 5, 10, 14]], clw=np.zeros(4), irt=3, dvset=dv))
        return prs

    prs = profiles(0.1)
    note = (f"line file shaped like an aer_v_3.x product: {len(rec)} records in 23 blocks, second header record ('^'), short blocks, "
            f"isotopologues 1-5, O2 60-GHz complex with coupling pairs, lines of molecules 9-12 beyond NMOL; the coupling record of the "
            f"118.75 GHz line is the first record of a block (record {info['lc_slot1']}); sounder channels down / up (IBRD = 1) and a 1200-point grid")

    def save_kw(name, nt):
        # (save() writes the file itself: pass the layout through)
        import tempfile as tf
        with tf.TemporaryDirectory() as d:
            tp, cp, op = (os.path.join(d, x) for x in ("TAPE3", "case.bin", "out.bin"))
            tape3.write_tape3(tp, rec, **kw)
            caseio.write_case(cp, prs)
            r = subprocess.run([HARNESS, cp, tp, op], cwd=d, capture_output=True, text=True)
            if r.returncode != 0 or "HARNESS_SECONDS" not in r.stdout:
                raise RuntimeError(f"reference harness failed: rc={r.returncode}\n{r.stdout[-2000:]}\n{r.stderr[-2000:]}")
            dumps = caseio.read_dump(op)
            out = dict(tape3=np.frombuffer(open(tp, "rb").read(), np.uint8), case=np.frombuffer(open(cp, "rb").read(), np.uint8),
                       nprof=np.int32(len(dumps)), note=np.array(nt))
        for k, dmp in enumerate(dumps):
            for f in ("o", "o_by_mol", "oc", "o_clw", "rup", "rdn", "trtot", "rad", "tb", "tmr"):
                v = getattr(dmp, f)
                if name.startswith("sgl_"):   # the REAL*4 values of the sgl build, stored as such (exact)
                    assert np.array_equal(v.astype(np.float32).astype(np.float64), v, equal_nan=True), (name, f)
                    v = v.astype(np.float32)
                out[f"p{k}_{f}"] = v
            out[f"p{k}_tmpsfc_out"] = np.float64(dmp.tmpsfc_out)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        fin = all(np.isfinite(getattr(d, f)).all() for d in dumps for f in ("o_by_mol", "tb"))
        print(f"{name:28s} {os.path.getsize(path)/1024:8.1f} KiB  nprof={len(dumps)} finite={fin} TB[{dumps[0].tb.min():.2f},{dumps[0].tb.max():.2f}] "
              f"O2 od range [{dumps[0].o_by_mol[:, 6].min():.3g}, {dumps[0].o_by_mol[:, 6].max():.3g}] N2O [{dumps[0].o_by_mol[:, 3].min():.3g}, {dumps[0].o_by_mol[:, 3].max():.3g}]")

    save_kw("real_like", note)
    keep = HARNESS
    HARNESS = os.path.join(ROOT, "oracle", "_ref", "harness_ref_sgl")
    # (with the model top at 0.1 hPa the REAL*4 build of SDVOIGT returns Re(v) < 0 for the speed-dependent 183 GHz line and the
    # reference STOPs, src/modm.f90:1062 - as on voigt_regions: the single-precision twin keeps to pressures above 20 hPa)
    prs = profiles(20.0)
    try:
        save_kw("sgl_real_like", "the line file of real_like, model top at 20 hPa, outputs of the SINGLE-PRECISION reference build")
    finally:
        HARNESS = keep


def gen_dense_far():
    """Dense grids of five tiles of 512 wavenumbers - the shape on which the HIP path forms the far field of every tile outside its
    line kernel, in levels of intervals (far_kernel.hip) - made by BOTH reference builds: 2100 points, 0.004 cm-1 apart, two layers,
    2500 lines of the usual molecule mix incl. CO2 (quadratic pedestal), O2 (no pedestal) and two-resonance lines below 25 cm-1; the
    single-precision twin on the sgl driver's own REAL*4 grid with 400 lines (short REAL*4 sums: held to 2e-4)."""
    global HARNESS
    n = 2100
    a = synth.standard_atmosphere(2, ztop_km=12)
    rec = synth.synthetic_lines(2500, seed=515, vlo=0.05, vhi=54.9)
    wn = 6.0 + 0.004 * np.arange(n)
    pr = synth.Profile(wn=wn, p=a["p"], t=a["t"], tz=a["tz"], wkl=a["wkl"], wbrodl=a["wbrodl"], clw=a["clw"], irt=3, dvset=0.004)
    save("dense_far", rec, [pr], note="dense grid: 2100 points from 6 cm-1, DVSET = 0.004, two layers, 2500 lines (far field of five tiles)")
    keep = HARNESS
    HARNESS = os.path.join(ROOT, "oracle", "_ref", "harness_ref_sgl")
    try:
        dv4 = np.float32(0.004)
        wn4 = 6.0 + (np.arange(n, dtype=np.float32) * dv4).astype(np.float64)   # REAL*4 product, REAL*8 sum (src/monortm_sub.F90:287)
        rec4 = synth.synthetic_lines(400, seed=516, vlo=0.05, vhi=54.9)
        pr4 = synth.Profile(wn=wn4, p=a["p"], t=a["t"], tz=a["tz"], wkl=a["wkl"], wbrodl=a["wbrodl"], clw=a["clw"], irt=3, dvset=float(dv4))
        save("sgl_dense_far", rec4, [pr4], note="SINGLE-PRECISION reference build: dense grid of 2100 points on the sgl driver's REAL*4 grid, 400 lines")
    finally:
        HARNESS = keep


ALL = [gen_dense_far, gen_real_like, gen_negative_nan, gen_sgl_grid, gen_xsec, gen_all_molecules, gen_self_coupling, gen_ir_uv, gen_sgl_cloud, gen_sgl_more, gen_c2, gen_c2_lc_sdep, gen_voigt, gen_lc, gen_cloud_up, gen_grid_ir, gen_cntnm_factors, gen_ibrd,
       gen_cut_boundaries, gen_temperature_brackets]

if __name__ == "__main__":
    if not os.path.exists(HARNESS):
        sys.exit("oracle/_ref/harness_ref_dbl missing: run `make -C oracle ref` where /root/reference exists")
    sel = sys.argv[1:]
    for g in ALL:
        if not sel or g.__name__ in sel:
            g()
