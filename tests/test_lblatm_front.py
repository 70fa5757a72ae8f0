"""f2 - the own IATM = 1 layering front end (monortm_amd/fortran/lblatm_front.f90) against the reference's LBLATM.

CPU part (no GPU): the driver's MONORTM_LAYERS_ONLY mode writes the layer quantities it would hand to the GPU; they must
equal the reference's TAPE7 for the model-atmosphere decks (example cases 1 and 2: U.S. standard atmosphere, vertical
path 0-30 km looking up / 30-0 km looking down, automatic layering; cases 3 and 6: user profile with unit keys on 61 given
boundaries; case 7: 83-level user profile with 19 molecules seen from 120 km, automatic layering) to the precision TAPE7 prints: pressures and column
amounts to 8 significant digits (tolerance 3e-7), temperatures to 0.01 K.  Same layer count, same boundaries.
GPU part: MONORTM.OUT of the own driver on those decks against the reference program's (tests/test_reference_driver_dropin)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from common import ROOT

DECKS = os.path.join(ROOT, "tests", "golden", "decks")
CASES = [c for c in sorted(os.listdir(DECKS)) if os.path.exists(os.path.join(DECKS, c, "TAPE7.expected"))]


def read_tape7(path):
    L = open(path).read().split("\n")
    nlay, nmol = int(L[1][2:5]), int(L[1][5:10])
    ang = float(L[1].split("ANG=")[1][:8])
    pos, out = 2, []
    for lay in range(nlay):
        ln = L[pos]
        pos += 1
        rec = {"p": float(ln[0:15]), "t": float(ln[15:25])}
        if lay == 0:
            rec["tz0"], rec["tz"] = float(ln[56:63]), float(ln[78:85])
        else:
            rec["tz"] = float(ln[78:85])
        first = [float(L[pos][i * 15:(i + 1) * 15]) for i in range(8)]
        pos += 1
        wk, rec["wb"] = first[:7], first[7]
        rest = nmol - 7
        while rest > 0:
            n = min(8, rest)
            wk += [float(L[pos][i * 15:(i + 1) * 15]) for i in range(n)]
            pos += 1
            rest -= n
        rec["wk"] = np.array(wk)
        out.append(rec)
    return nlay, nmol, ang, out


def read_layers(path):
    L = open(path).read().split("\n")
    _, nlay, nmol = (int(x) for x in L[0].split()[:3])
    ang = float(L[0].split()[3])
    pos, out = 1, []
    for _ in range(nlay):
        a = L[pos].split()
        pos += 1
        vals = []
        while len(vals) < nmol + 1:
            vals += [float(x) for x in L[pos].split()]
            pos += 1
        out.append({"p": float(a[1]), "t": float(a[2]), "tz0": float(a[3]), "tz": float(a[4]), "wk": np.array(vals[:nmol]), "wb": vals[nmol]})
    return nlay, nmol, ang, out


FUZZ = os.path.join(ROOT, "tests", "golden", "layers_fuzz")
FUZZ_CASES = sorted(os.listdir(FUZZ)) if os.path.isdir(FUZZ) else []


def _compare_layers(deck_dir, tmp_path, nlay_expected=None):
    from monortm_amd import _build

    exe = _build.build_fortran_shim()["driver"]
    shutil.copy(os.path.join(deck_dir, "MONORTM.IN"), tmp_path)
    r = subprocess.run([exe], cwd=tmp_path, capture_output=True, text=True, timeout=120, env={**os.environ, "MONORTM_LAYERS_ONLY": "1"})
    assert "LAYERS.OUT written" in r.stdout, (r.stdout + r.stderr)[-2000:]
    nl_r, nm_r, ang_r, ref = read_tape7(os.path.join(deck_dir, "TAPE7.expected"))
    nl_o, nm_o, ang_o, own = read_layers(tmp_path / "LAYERS.OUT")
    assert (nl_o, nm_o) == (nl_r, nm_r) and abs(ang_o - ang_r) <= 5.1e-4  # F8.3 in TAPE7
    if nlay_expected is not None:
        assert nl_r == nlay_expected
    return ref, own


@pytest.mark.parametrize("case", FUZZ_CASES)
def test_layers_match_reference_random_decks(case, tmp_path):
    """24 seeded random decks (tests/golden/make_layer_fuzz.py): the six model atmospheres, paths 2A / 3A / 3B, automatic
    layering with various AVTRAT / TDIFF, given altitude or pressure boundaries, 7-28 molecules, NOZERO on and off."""
    ref, own = _compare_layers(os.path.join(FUZZ, case), tmp_path)
    for lay, (a, b) in enumerate(zip(ref, own)):
        assert abs(a["p"] - b["p"]) <= 6e-7 * a["p"], (lay, a["p"], b["p"])
        assert abs(a["t"] - b["t"]) <= 0.0051 and abs(a["tz"] - b["tz"]) <= 0.0051, (lay, a, b)
        assert abs(a["wb"] - b["wb"]) <= 3e-7 * abs(a["wb"]) + 1.0, (lay, a["wb"], b["wb"])
        err = np.abs(a["wk"] - b["wk"]) / np.maximum(np.abs(a["wk"]), 1e-300)
        err[(a["wk"] == 0) & (b["wk"] == 0)] = 0
        assert err.max() <= 3e-7, (lay, int(np.argmax(err)) + 1, err.max(), a["wk"], b["wk"])


@pytest.mark.parametrize("case", CASES)
def test_layers_match_reference_tape7(case, tmp_path):
    nlay = {"case1": 35, "case2": 35, "case3": 60, "case6": 60, "case7": 66, "case11": 8, "case12": 19, "case13": 62}[case.split("_")[0]]
    ref, own = _compare_layers(os.path.join(DECKS, case), tmp_path, nlay)
    for lay, (a, b) in enumerate(zip(ref, own)):
        assert abs(a["p"] - b["p"]) <= 6e-7 * a["p"], (lay, a["p"], b["p"])             # printed with 7 significant digits
        assert abs(a["t"] - b["t"]) <= 0.0051, (lay, a["t"], b["t"])                    # printed F10.2
        assert abs(a["tz"] - b["tz"]) <= 0.0051 and (lay > 0 or abs(a["tz0"] - b["tz0"]) <= 0.0051)
        assert abs(a["wb"] - b["wb"]) <= 3e-7 * abs(a["wb"])
        err = np.abs(a["wk"] - b["wk"]) / np.maximum(np.abs(a["wk"]), 1e-300)
        assert err.max() <= 3e-7, (lay, int(np.argmax(err)) + 1, err.max())                # 1P8E15.7


def test_front_end_refuses_what_it_does_not_cover(tmp_path):
    """Options outside the built front end (here: a horizontal path, ITYPE = 1) are not silently mis-handled: the driver
    stops with a message."""
    from monortm_amd import _build

    exe = _build.build_fortran_shim()["driver"]
    lines = open(os.path.join(DECKS, "case1_MDL_ATM_dn", "MONORTM.IN")).read().split("\n")
    k = next(i for i, ln in enumerate(lines) if ln.startswith("    6    2    0"))      # record 3.1: MODEL, ITYPE, IBMAX
    lines[k] = "    6    1" + lines[k][10:]
    open(tmp_path / "MONORTM.IN", "w").write("\n".join(lines))
    r = subprocess.run([exe], cwd=tmp_path, capture_output=True, text=True, timeout=120, env={**os.environ, "MONORTM_LAYERS_ONLY": "1"})
    assert r.returncode != 0 and "ITYPE must be 2 or 3" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_own_driver_iatm1_model_atmospheres(case, tmp_path):
    """End to end without any reference code: MONORTM.IN (IATM = 1) -> own layering -> GPU -> MONORTM.OUT, against the
    reference program's MONORTM.OUT on the same deck."""
    from monortm_amd import _build
    from test_reference_driver_dropin import check_out

    exe = _build.build_fortran_shim()["driver"]
    shutil.copy(os.path.join(DECKS, case, "MONORTM.IN"), tmp_path)
    shutil.copy(os.path.join(DECKS, "TAPE3_synthetic"), tmp_path / "TAPE3")
    r = subprocess.run([exe], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    check_out(tmp_path / "MONORTM.OUT", os.path.join(DECKS, case, "MONORTM.OUT.expected"))
