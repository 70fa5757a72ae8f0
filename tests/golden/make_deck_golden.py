#!/usr/bin/env python3
"""File-interface fixtures: the reference PROGRAM MONORTM (oracle/_ref/monortm_ref_dbl) run on its own
example decks (run/in/*, run/run_monortm_examples cases 1-6 and the lidar deck) with a synthetic TAPE3 (the reference's
line file is a dangling symlink).  Stores inputs + the reference's MONORTM.OUT under tests/golden/decks/.
Only runs where /root/reference and the compiled reference exist."""
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from monortm_amd import synth, tape3  # noqa: E402

REF_IN = "/root/reference/run/in"
EXE = os.path.join(ROOT, "oracle", "_ref", "monortm_ref_dbl")
OUT = os.path.join(ROOT, "tests", "golden", "decks")

CASES = {  # name: (MONORTM.IN deck, MONORTM_PROF.IN or None)   -- run/run_monortm_examples:18-110
    "case1_MDL_ATM_dn": ("MONORTM.IN_MDL_ATM_dn", None),
    "case2_MDL_ATM_up": ("MONORTM.IN_MDL_ATM_up", None),
    "case3_NOSCALE_IATM1_dn": ("MONORTM.IN_NOSCALE_IATM1_dn", None),
    "case6_SCALE_IATM1_MODEL0_HMOL1_dn": ("MONORTM.IN_SCALE_IATM1_MODEL0_HMOL1_dn", None),
    "case7_IATM1_lidar_up": ("MONORTM.IN_IATM1_lidar_up", None),
    "case4_IATM0_dn": ("MONORTM.IN_IATM0_dn", "MONORTM_PROF.IN_sav"),
    "case5_IATM0_liquid_cloud": ("MONORTM.IN_IATM0_dn", "MONORTM_PROF.IN_liquid_cloud"),
    # three concatenated profiles (the reference counts them in GETPROFNUMBER, src/monortm_sub.F90:895-900)
    "case45_IATM0_three_profiles": ("MONORTM.IN_IATM0_dn", ("MONORTM_PROF.IN_sav", "MONORTM_PROF.IN_liquid_cloud",
                                                             "MONORTM_PROF.IN_sav")),
}

def with_iod(text: str) -> str:
    """Record 1.2 (format 925, src/monortm_sub.F90:402): IOD is the I1 in column 65 - the line after the '$' record."""
    lines = text.split("\n")
    k = next(i for i, ln in enumerate(lines) if ln.startswith("$")) + 1
    ln = lines[k].ljust(65)
    lines[k] = ln[:64] + "1" + ln[65:]
    return "\n".join(lines)


IOD_CASES = {"case8_IATM0_IOD1_layer_od": ("MONORTM.IN_IATM0_dn", ("MONORTM_PROF.IN_sav", "MONORTM_PROF.IN_liquid_cloud"))}

if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    t3 = os.path.join(OUT, "TAPE3_synthetic")
    # molecules <= 7 only (HALFWHM_C reads out of bounds for others), half of the O2 lines coupled
    if not os.path.exists(t3) or "--all" in sys.argv:
        tape3.write_tape3(t3, synth.synthetic_lines(300, seed=77, vlo=0.05, vhi=40.0, lc_frac=0.5))
    for name, (deck, prof) in CASES.items():
        d = os.path.join(OUT, name)
        if os.path.exists(os.path.join(d, "MONORTM.OUT.expected")) and "--all" not in sys.argv:
            continue  # fixtures already committed: keep them byte-identical
        os.makedirs(d, exist_ok=True)
        shutil.copy(os.path.join(REF_IN, deck), os.path.join(d, "MONORTM.IN"))
        if isinstance(prof, tuple):
            with open(os.path.join(d, "MONORTM_PROF.IN"), "w") as f:
                for q in prof:
                    f.write(open(os.path.join(REF_IN, q)).read())
        elif prof:
            shutil.copy(os.path.join(REF_IN, prof), os.path.join(d, "MONORTM_PROF.IN"))
        with tempfile.TemporaryDirectory() as w:
            for f in os.listdir(d):
                if f.startswith("MONORTM") and f.endswith(".IN"):
                    shutil.copy(os.path.join(d, f), w)
            shutil.copy(t3, os.path.join(w, "TAPE3"))
            r = subprocess.run([EXE], cwd=w, capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stdout[-2000:]
            shutil.copy(os.path.join(w, "MONORTM.OUT"), os.path.join(d, "MONORTM.OUT.expected"))
        os.chmod(os.path.join(d, "MONORTM.IN"), 0o644)
        print(name, "ok")
    # IOD = 1: the reference also writes ODmono_prfNNNN_layNNNN (src/monortm_sub.F90:677-694); kept as ONE text file
    # (name line + content per file) next to MONORTM.OUT.expected
    t3 = os.path.join(OUT, "TAPE3_synthetic")
    for name, (deck, profs) in IOD_CASES.items():
        d = os.path.join(OUT, name)
        if os.path.exists(os.path.join(d, "ODmono.expected")) and "--all" not in sys.argv:
            continue
        os.makedirs(d, exist_ok=True)
        open(os.path.join(d, "MONORTM.IN"), "w").write(with_iod(open(os.path.join(REF_IN, deck)).read()))
        with open(os.path.join(d, "MONORTM_PROF.IN"), "w") as f:
            for q in profs:
                f.write(open(os.path.join(REF_IN, q)).read())
        with tempfile.TemporaryDirectory() as w:
            for f in ("MONORTM.IN", "MONORTM_PROF.IN"):
                shutil.copy(os.path.join(d, f), w)
            shutil.copy(t3, os.path.join(w, "TAPE3"))
            r = subprocess.run([EXE], cwd=w, capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stdout[-2000:]
            shutil.copy(os.path.join(w, "MONORTM.OUT"), os.path.join(d, "MONORTM.OUT.expected"))
            names = sorted(f for f in os.listdir(w) if f.startswith("ODmono_prf"))
            assert names, "the reference wrote no ODmono files"
            with open(os.path.join(d, "ODmono.expected"), "w") as f:
                for n in names:
                    f.write(f"### {n}\n" + open(os.path.join(w, n)).read())
        print(name, "ok", len(names), "layer files")

    # f1 remainder: tabulated boundary emissivity / reflectivity (in/EMISSION, in/REFLECTION; src/monortm_sub.F90:1-29,
    # :317-335, :426-491) with an upwelling view, and profile scaling (records 1.3.a/b, src/monortm_sub.F90:937-1046) over
    # two profiles - the second profile sees the scale factors the first one left behind, as in the reference
    def deck_lines():
        return open(os.path.join(REF_IN, "MONORTM.IN_IATM0_dn")).read().split("\n")

    def k13(lines):
        return next(i for i, ln in enumerate(lines) if ln.startswith("$")) + 2

    def run_case(d, extra_in=()):
        with tempfile.TemporaryDirectory() as w:
            for f in ("MONORTM.IN", "MONORTM_PROF.IN"):
                shutil.copy(os.path.join(d, f), w)
            if os.path.isdir(os.path.join(d, "in")):
                shutil.copytree(os.path.join(d, "in"), os.path.join(w, "in"))
            shutil.copy(t3, os.path.join(w, "TAPE3"))
            r = subprocess.run([EXE], cwd=w, capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stdout[-2000:]
            shutil.copy(os.path.join(w, "MONORTM.OUT"), os.path.join(d, "MONORTM.OUT.expected"))

    d = os.path.join(OUT, "case9_IATM0_emis_refl_files_up")
    if not os.path.exists(os.path.join(d, "MONORTM.OUT.expected")) or "--all" in sys.argv:
        os.makedirs(os.path.join(d, "in"), exist_ok=True)
        lines = deck_lines()
        k = k13(lines) + 6                       # record 1.3, NWN, 4 wavenumbers -> record 1.4
        assert lines[k].split()[:2] == ["0.", "1.0"], lines[k]
        lines[k] = "".join(f"{v:10.3E}" for v in (290.0, -1.0, 0.0, 0.0, -1.0, 0.0, 0.0))
        open(os.path.join(d, "MONORTM.IN"), "w").write("\n".join(lines))
        prof = open(os.path.join(REF_IN, "MONORTM_PROF.IN_sav")).read() + open(os.path.join(REF_IN, "MONORTM_PROF.IN_liquid_cloud")).read()
        assert prof.count("ANG=   0.000") == 2
        open(os.path.join(d, "MONORTM_PROF.IN"), "w").write(prof.replace("ANG=   0.000", "ANG= 180.000"))
        for name, f0 in (("EMISSION", 0.55), ("REFLECTION", 0.40)):
            with open(os.path.join(d, "in", name), "w") as f:
                f.write(f"{0.0:10.3E}{2.0:10.3E}{0.1:10.3E}     {20:5d}\n")
                for i in range(20):
                    f.write(f"{f0 + 0.013 * i - 0.0004 * i * i:15.7E}\n")
        run_case(d)
        print("case9 ok")

    d = os.path.join(OUT, "case10_IATM0_nmol_scal")
    if not os.path.exists(os.path.join(d, "MONORTM.OUT.expected")) or "--all" in sys.argv:
        os.makedirs(d, exist_ok=True)
        lines = deck_lines()
        k = k13(lines)
        lines[k] = lines[k][:100].ljust(100) + f"{4:5d}"
        lines.insert(k + 1, "1MDC")                                                   # record 1.3.a: factor, mixing ratio, Dobson, column
        lines.insert(k + 2, "".join(f"{v:15.7E}" for v in (1.25, 4.1e-4, 310.0, 6.0e18)))   # record 1.3.b
        open(os.path.join(d, "MONORTM.IN"), "w").write("\n".join(lines))
        with open(os.path.join(d, "MONORTM_PROF.IN"), "w") as f:
            f.write(open(os.path.join(REF_IN, "MONORTM_PROF.IN_sav")).read() + open(os.path.join(REF_IN, "MONORTM_PROF.IN_liquid_cloud")).read())
        run_case(d)
        print("case10 ok")

    # f2, pressure-level inputs (IBMAX < 0: boundaries, H1 and H2 as pressures; IMMAX < 0: user profile on pressure levels, altitudes
    # from the hydrostatic equation CMPALT, model defaults through DEFALT_P).  case11: the U.S. standard atmosphere seen from
    # 12 mb down to 1013 mb on nine pressure boundaries; case12: the 2413 sonde levels of case 6 with their altitudes blanked
    # (only the first one counts) and 20 pressure boundaries
    d = os.path.join(OUT, "case11_MDL_ATM_pressure_boundaries_up")
    if not os.path.exists(os.path.join(d, "MONORTM.OUT.expected")) or "--all" in sys.argv:
        os.makedirs(d, exist_ok=True)
        lines = open(os.path.join(REF_IN, "MONORTM.IN_MDL_ATM_up")).read().split("\n")
        k = next(i for i, ln in enumerate(lines) if ln.split()[:7] == ["6", "2", "0", "1", "1", "22", "1"])
        pb = [1013.0, 950.0, 850.0, 700.0, 500.0, 300.0, 100.0, 50.0, 12.0]
        lines[k] = f"{6:5d}{2:5d}{-len(pb):5d}{1:5d}{1:5d}{22:5d}{1:5d}"
        lines[k + 1] = f"{12.0:10.3f}{1013.0:10.3f}{180.0:10.3f}"
        lines[k + 2] = "".join(f"{v:10.3f}" for v in pb[:8]) + "\n" + "".join(f"{v:10.3f}" for v in pb[8:])
        open(os.path.join(d, "MONORTM.IN"), "w").write("\n".join(lines))
    d = os.path.join(OUT, "case12_MODEL0_pressure_levels_dn")
    if not os.path.exists(os.path.join(d, "MONORTM.OUT.expected")) or "--all" in sys.argv:
        os.makedirs(d, exist_ok=True)
        lines = open(os.path.join(REF_IN, "MONORTM.IN_SCALE_IATM1_MODEL0_HMOL1_dn")).read().split("\n")
        k = next(i for i, ln in enumerate(lines) if ln.split()[:7] == ["0", "2", "61", "1", "0", "7", "1"])
        pb = [1011.9, 1000.0, 975.0, 950.0, 925.0, 900.0, 850.0, 800.0, 700.0, 600.0, 500.0, 400.0, 300.0, 250.0, 200.0, 150.0, 100.0,
              70.0, 50.0, 30.0]
        lines[k] = f"{0:5d}{2:5d}{-len(pb):5d}{1:5d}{0:5d}{7:5d}{1:5d}"
        lines[k + 1] = f"{pb[0]:10.3f}{pb[-1]:10.3f}{0.0:10.3f}"
        nb = 8                                  # the 61 altitudes take 8 lines of 8
        assert lines[k + 2 + nb].split()[0] == "2418", lines[k + 2 + nb]
        bl = ["".join(f"{v:10.3f}" for v in pb[i:i + 8]) for i in range(0, len(pb), 8)]
        lines[k + 2:k + 2 + nb] = bl
        k4 = k + 2 + len(bl)
        # the five levels above the sonde (model defaults for P and T at given altitudes) cannot be stated on pressure levels
        nlev = 2413
        assert lines[k4 + 1 + 2 * nlev][30:40].split() == ["66"], lines[k4 + 1 + 2 * nlev]
        del lines[k4 + 1 + 2 * nlev:k4 + 1 + 2 * 2418]
        lines[k4] = f"{-nlev:5d}" + lines[k4][5:]
        j = k4 + 1
        for lev in range(nlev):                 # record 3.5: the altitude field counts for the first level only
            assert lines[j][35:37] == "AA", lines[j]
            if lev > 0:
                lines[j] = f"{0.0:10.3f}" + lines[j][10:]
            j += 2
        open(os.path.join(d, "MONORTM.IN"), "w").write("\n".join(lines))
    # f2, limb geometry: FSCGEO case 3B (ITYPE = 3 with the tangent height in the H2 field): observer at 30 km looking through a
    # tangent height of 10 km to space, U.S. standard atmosphere, automatic layering
    d = os.path.join(OUT, "case13_MDL_ATM_limb_3B")
    if not os.path.exists(os.path.join(d, "MONORTM.OUT.expected")) or "--all" in sys.argv:
        os.makedirs(d, exist_ok=True)
        lines = open(os.path.join(REF_IN, "MONORTM.IN_MDL_ATM_up")).read().split("\n")
        k = next(i for i, ln in enumerate(lines) if ln.split()[:7] == ["6", "2", "0", "1", "1", "22", "1"])
        lines[k] = f"{6:5d}{3:5d}{0:5d}{1:5d}{1:5d}{22:5d}{1:5d}"
        lines[k + 1] = f"{30.0:10.3f}{10.0:10.3f}{0.0:10.3f}"
        open(os.path.join(d, "MONORTM.IN"), "w").write("\n".join(lines))
    for name in ("case11_MDL_ATM_pressure_boundaries_up", "case12_MODEL0_pressure_levels_dn", "case13_MDL_ATM_limb_3B"):
        d = os.path.join(OUT, name)
        if os.path.exists(os.path.join(d, "TAPE7.expected")) and "--all" not in sys.argv:
            continue
        with tempfile.TemporaryDirectory() as w:
            shutil.copy(os.path.join(d, "MONORTM.IN"), w)
            shutil.copy(t3, os.path.join(w, "TAPE3"))
            r = subprocess.run([EXE], cwd=w, capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stdout[-3000:]
            shutil.copy(os.path.join(w, "MONORTM.OUT"), os.path.join(d, "MONORTM.OUT.expected"))
            shutil.copy(os.path.join(w, "TAPE7"), os.path.join(d, "TAPE7.expected"))
        print(name, "ok")

    # f2: the layer quantities the reference's LBLATM hands to the hot path for the model-atmosphere decks (its TAPE7,
    # written because IPUNCH = 1 on record 3.1): fixtures for the own IATM = 1 front end (lblatm_front.f90)
    for name in ("case1_MDL_ATM_dn", "case2_MDL_ATM_up", "case3_NOSCALE_IATM1_dn", "case6_SCALE_IATM1_MODEL0_HMOL1_dn",
                 "case7_IATM1_lidar_up"):
        d = os.path.join(OUT, name)
        if os.path.exists(os.path.join(d, "TAPE7.expected")) and "--all" not in sys.argv:
            continue
        with tempfile.TemporaryDirectory() as w:
            shutil.copy(os.path.join(d, "MONORTM.IN"), w)
            shutil.copy(t3, os.path.join(w, "TAPE3"))
            r = subprocess.run([EXE], cwd=w, capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stdout[-2000:]
            assert open(os.path.join(w, "MONORTM.OUT")).read() == open(os.path.join(d, "MONORTM.OUT.expected")).read()
            shutil.copy(os.path.join(w, "TAPE7"), os.path.join(d, "TAPE7.expected"))
        print(name, "TAPE7 ok")
