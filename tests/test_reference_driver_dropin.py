"""End-to-end drop-in through the reference's OWN driver and file interface.

oracle/_ref/monortm_hipdrop_dbl is the reference's PROGRAM MONORTM, RDLBLINP, LBLATM and STOREOUT compiled
unchanged, with src/modm.f90 and src/RTMmono.f90 replaced by monortm_amd/fortran/*_hip.f90 (oracle/Makefile,
INTEGRATION.md section 1).  It reads MONORTM.IN / MONORTM_PROF.IN / TAPE3 and writes MONORTM.OUT; the expected
MONORTM.OUT files come from the unmodified reference program (tests/golden/make_deck_golden.py) on the
reference's own example decks (run/run_monortm_examples cases 1-6 and the lidar deck)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from common import ROOT

pytestmark = pytest.mark.gpu
DECKS = os.path.join(ROOT, "tests", "golden", "decks")
EXE = os.path.join(ROOT, "oracle", "_ref", "monortm_hipdrop_dbl")
CASES = sorted(d for d in os.listdir(DECKS) if os.path.isdir(os.path.join(DECKS, d)))


def parse_out(path):
    """-> list of rows of floats (one per wavenumber line of MONORTM.OUT, format 21 of
    src/monortm_sub.F90:781-782: i5 NPR, f10.3 FREQ, then blank-separated columns)."""
    rows = []
    for line in open(path):
        if len(line) > 150 and line[:5].strip().isdigit():
            try:
                rows.append([float(line[:5]), float(line[5:15])] + [float(t) for t in line[15:].split()])
            except ValueError:
                pass
    return np.array(rows)


def check_out(got_path, exp_path):
    got = parse_out(got_path)
    exp = parse_out(exp_path)
    assert got.shape == exp.shape and got.shape[0] > 0
    # columns: NPR FREQ BT TMR RAD TRANS PWV CLW TBOUND EMIS REFL ANGLE TOTAL_OD per-molecule ODs ...
    # printed precision: BT/TMR f11.5, RAD 1p E21.9 (10 digits), TRANS f9.5, ODs 1p E12.4 (5 digits)
    assert np.allclose(got[:, 2:4], exp[:, 2:4], rtol=1e-6, atol=2e-5), "BT / TMR"
    assert np.allclose(got[:, 4], exp[:, 4], rtol=1e-6, atol=0), "RAD"
    assert np.allclose(got[:, 5], exp[:, 5], rtol=0, atol=1.1e-5), "TRANS"
    assert np.allclose(got[:, 12:], exp[:, 12:], rtol=2e-4, atol=1e-30), "optical depths (5 printed digits)"
    assert np.array_equal(got[:, 0], exp[:, 0]) and np.allclose(got[:, 6:12], exp[:, 6:12], rtol=0, atol=1e-4)
    # FREQ: the reference's writer prints GHz when wn(1) < 100 and otherwise leaves its LOGICAL `giga` uninitialised
    # (src/monortm_sub.F90:622-628), so above 100 cm-1 either unit can come out of the reference's own STOREOUT
    ghz = 2.99792458e10 / 1.0e9
    same = np.allclose(got[:, 1], exp[:, 1], rtol=0, atol=1.1e-3)
    other = exp[0, 1] >= 100. and (np.allclose(got[:, 1], exp[:, 1] * ghz, rtol=1e-6) or np.allclose(got[:, 1] * ghz, exp[:, 1], rtol=1e-6))
    assert same or other, "FREQ"


def stage_inputs(src, dst):
    """MONORTM.IN, MONORTM_PROF.IN and, where the deck asks for tabulated boundary properties, in/EMISSION / in/REFLECTION."""
    for f in os.listdir(src):
        if f.endswith(".IN"):
            shutil.copy(os.path.join(src, f), dst)
    if os.path.isdir(os.path.join(src, "in")):
        shutil.copytree(os.path.join(src, "in"), os.path.join(dst, "in"))


def check_layer_od(run_dir, case_dir):
    """IOD = 1: ODmono_prfNNNN_layNNNN files (src/monortm_sub.F90:677-694) against the reference program's, same file
    set, same text layout, optical depths to their 4 printed digits."""
    exp_path = os.path.join(case_dir, "ODmono.expected")
    if not os.path.exists(exp_path):
        assert not [f for f in os.listdir(run_dir) if f.startswith("ODmono_prf")]
        return 0
    blocks = open(exp_path).read().split("### ")[1:]
    names = sorted(f for f in os.listdir(run_dir) if f.startswith("ODmono_prf"))
    assert names == [b.split("\n", 1)[0] for b in blocks]
    for b in blocks:
        name, body = b.split("\n", 1)
        got = open(os.path.join(run_dir, name)).read().splitlines()
        exp = body.splitlines()
        assert len(got) == len(exp) and got[0] == exp[0] and got[1] == exp[1], name
        for a, e in zip(got[2:], exp[2:]):
            assert len(a) == len(e) == 22 and a[:10] == e[:10], (name, a, e)
            assert abs(float(a[10:]) - float(e[10:])) <= 2e-4 * abs(float(e[10:])) + 1e-30, (name, a, e)
    return len(blocks)


@pytest.mark.parametrize("case", [c for c in CASES if "IATM0" in c])
def test_own_driver_iatm0(case, tmp_path):
    """monortm_amd/fortran/monortm_driver.f90: our own MONORTM.IN / MONORTM_PROF.IN / MONORTM.OUT driver (no reference
    code), one batched GPU call for all profiles of the file."""
    from monortm_amd import _build

    exe = _build.build_fortran_shim()["driver"]
    src = os.path.join(DECKS, case)
    stage_inputs(src, tmp_path)
    shutil.copy(os.path.join(DECKS, "TAPE3_synthetic"), tmp_path / "TAPE3")
    r = subprocess.run([exe], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    check_out(tmp_path / "MONORTM.OUT", os.path.join(src, "MONORTM.OUT.expected"))
    # text layout identical to the reference's writer: same header lines, same column count per row
    assert check_layer_od(tmp_path, src) == (38 if "IOD1" in case else 0)
    got_lines = open(tmp_path / "MONORTM.OUT").read().splitlines()
    exp_lines = open(os.path.join(src, "MONORTM.OUT.expected")).read().splitlines()
    assert len(got_lines) == len(exp_lines)
    for a, b in zip(got_lines, exp_lines):
        assert len(a.rstrip()) == len(b.rstrip())
        if not a[:5].strip().isdigit():
            assert a.rstrip() == b.rstrip()


@pytest.mark.parametrize("case", CASES)
def test_reference_driver_with_hip_modules(case, tmp_path):
    if not os.path.exists(EXE):
        pytest.skip("oracle/_ref/monortm_hipdrop_dbl not built (needs the reference tree: make -C oracle ref)")
    src = os.path.join(DECKS, case)
    stage_inputs(src, tmp_path)
    shutil.copy(os.path.join(DECKS, "TAPE3_synthetic"), tmp_path / "TAPE3")
    r = subprocess.run([EXE], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    check_out(tmp_path / "MONORTM.OUT", os.path.join(src, "MONORTM.OUT.expected"))
    check_layer_od(tmp_path, src)
