"""IXSECT = 1 through the file interface: the own driver (monortm_amd/fortran/monortm_driver.f90) reads records 2.2.x of
MONORTM_PROF.IN, FSCDXS and the xs files, and writes the XSEC_OD column of MONORTM.OUT.  The reference PROGRAM cannot
supply the expected file: with the XSCT flag set it dies with SIGSEGV on this very deck (tools/xsec_reference_deck.py; its
driver hands MONORTM_XSEC_SUB an ODXSEC(nwn, .) that the routine indexes as (NWNMX, MXLAY), src/monortm_sub.F90:1611).  The
column is therefore held to the oracle's restatement of MONORTM_XSEC_SUB (pinned to the compiled routine by the xsec_*
fixtures) on the layer values of the deck, and to the difference of the total optical depth with and without the flag."""
import importlib.util
import os
import subprocess

import numpy as np
import pytest

from common import ROOT

pytestmark = pytest.mark.gpu


def _tool():
    spec = importlib.util.spec_from_file_location("xsec_reference_deck", os.path.join(ROOT, "tools", "xsec_reference_deck.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _columns(path):
    rows = []
    for ln in open(path).read().split("\n")[4:]:
        w = ln.split()
        if len(w) > 13:
            rows.append([float(x) for x in w])
    return np.array(rows)


def test_own_driver_writes_the_cross_section_column(tmp_path):
    from monortm_amd import _build, xsec
    from oracle import pyoracle

    exe = _build.build_fortran_shim()["driver"]
    tool = _tool()
    outs = {}
    for flag in (1, 0):
        d = str(tmp_path / f"xs{flag}")
        wn, P, T, xamnt, names = tool.build_deck(d, ixsect=flag)
        r = subprocess.run([exe], cwd=d, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and os.path.exists(os.path.join(d, "MONORTM.OUT")), (r.stdout + r.stderr)[-2000:]
        outs[flag] = _columns(os.path.join(d, "MONORTM.OUT"))
        assert len(outs[flag]) == len(wn)
    xs_od, tot1, tot0 = outs[1][:, -1], outs[1][:, 12], outs[0][:, 12]
    assert np.all(outs[0][:, -1] == 0.0) and np.all(xs_od > 0)
    # the oracle's MONORTM_XSEC_SUB on the deck's layers (printed with 8 / 5 significant digits)
    tabs = xsec.load_tables(str(tmp_path / "xs1"), names, float(wn.min()), float(wn.max()))
    reg, temps, pres, offs, pool = tabs.flatten()
    odx = np.zeros((len(P), len(wn)))
    L = pyoracle.lib()
    L.orc_xsec(len(wn), np.ascontiguousarray(wn), len(P), np.ascontiguousarray(P), np.ascontiguousarray(T), len(names), len(reg),
               np.ascontiguousarray(reg), np.ascontiguousarray(temps), np.ascontiguousarray(pres), np.ascontiguousarray(offs), pool,
               np.ascontiguousarray(xamnt), odx)
    want = odx.sum(axis=0)
    assert np.allclose(xs_od, want, rtol=2e-4), (xs_od, want)          # E12.4 column, F10.4 temperatures in the deck
    assert np.allclose(tot1 - tot0, xs_od, rtol=2e-3, atol=1e-4 * tot1.max())   # ODXSEC enters O (src/modm.f90:268)
    assert np.all(outs[1][:, 2] != outs[0][:, 2])                      # ... and through it the brightness temperature


def test_reference_driver_with_dropin_modm_handles_cross_sections(tmp_path):
    """The reference's own PROGRAM MONORTM, XSREAD and input layer, relinked against the drop-in modules (oracle/_ref/
    monortm_hipdrop_dbl): with MODM replaced, the IXSECT = 1 deck that kills the reference runs through - the shim indexes
    the caller's ODXSEC as the caller dimensioned it - and MONORTM.OUT equals the own driver's, XSEC_OD column included."""
    from monortm_amd import _build

    drop = os.path.join(ROOT, "oracle", "_ref", "monortm_hipdrop_dbl")
    if not os.path.exists(drop):
        pytest.skip("oracle/_ref/monortm_hipdrop_dbl not built (needs the reference tree: make -C oracle ref)")
    own = _build.build_fortran_shim()["driver"]
    tool = _tool()
    cols = []
    for exe, tag in ((drop, "drop"), (own, "own")):
        d = str(tmp_path / tag)
        tool.build_deck(d, ixsect=1)
        r = subprocess.run([exe], cwd=d, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and os.path.getsize(os.path.join(d, "MONORTM.OUT")) > 0, (r.stdout + r.stderr)[-2000:]
        cols.append(_columns(os.path.join(d, "MONORTM.OUT")))
    assert cols[0].shape == cols[1].shape and np.all(cols[0][:, -1] > 0)
    # (column 1, the frequency: the reference's driver decides GHz / cm-1 with a LOGICAL it only sets when wn(1) < 100,
    # src/monortm_sub.F90:549,:623 - left out)
    keep = [c for c in range(cols[0].shape[1]) if c != 1]
    assert np.allclose(cols[0][:, keep], cols[1][:, keep], rtol=1e-4, atol=1e-6)
