"""Shared helpers for the parity tests."""
from __future__ import annotations

import glob
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from monortm_amd import caseio, synth  # noqa: E402

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
FIELDS = ("o", "o_by_mol", "oc", "o_clw", "rup", "rdn", "trtot", "rad", "tb", "tmr")

# north_star tolerance: brightness temperature, radiance and layer optical depths within 1e-6
# relative of the double-precision reference.
RTOL = 1e-6

# isotopologues per molecule as TIPS_2003 knows them (typed from src/tips_2003.f90:361-369, NOT taken from the product's
# generated table header: the tests that use it guard that header)
TIPS_ISONM = [6, 9, 18, 5, 6, 3, 3, 3, 2, 1, 2, 1, 3, 1, 2, 2, 1, 2, 5, 3, 2, 1, 3, 2, 1, 2, 1, 1, 1, 1, 3, 1, 1, 1, 2, 1, 2, 2, 1]
# molecules whose lines the reference cannot evaluate: HALFWHM_C reads rho_molec(mol) beyond the 7-element array
# (src/modm.f90:845) and for these two the slot holds NaN bits in the -O0 flang build (NaN x 0), whatever the widths
ALLMOL_SKIP = (19, 20)


def golden_names(single_precision: bool = False):
    """Double-precision reference fixtures by default; the `sgl_*` ones come from the reference's "sgl" build."""
    names = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))
    # nan_*: fixtures whose reference outputs hold NaN (compare_nan_aware() below), tape3_*: line-file fixtures of the host parser
    names = [n for n in names if not n.startswith(("nan_", "tape3_"))]
    return [n for n in names if n.startswith("sgl_") == single_precision]


def read_case_bytes(buf: bytes) -> list[synth.Profile]:
    magic, nprof = struct.unpack_from("<ii", buf, 0)
    assert magic == caseio.MAGIC
    pos = 8
    out = []
    for _ in range(nprof):
        nwn, nlay, nmol, irt, iout, icp, ibrd, _ixs = struct.unpack_from("<8i", buf, pos)
        pos += 32
        sc = np.frombuffer(buf, np.float64, 12, pos)
        pos += 96

        def take(n):
            nonlocal pos
            a = np.frombuffer(buf, np.float64, n, pos).copy()
            pos += 8 * n
            return a

        wn, p, t, clw, wbrodl = take(nwn), take(nlay), take(nlay), take(nlay), take(nlay)
        tz = take(nlay + 1)
        wkl = take(nlay * nmol).reshape(nlay, nmol)
        emiss, reflc = take(nwn), take(nwn)
        xs_names = xamnt = None
        if _ixs == 1:
            (nxs,) = struct.unpack_from("<i", buf, pos)
            pos += 4
            xs_names = [buf[pos + 10 * k: pos + 10 * k + 10].decode("ascii").strip() for k in range(nxs)]
            pos += 10 * nxs
            xamnt = take(nlay * nxs).reshape(nlay, nxs)
        out.append(synth.Profile(xs_names=xs_names, xamnt=xamnt, wn=wn, p=p, t=t, tz=tz, wkl=wkl, wbrodl=wbrodl, clw=clw, irt=irt, tmpsfc=float(sc[4]),
                                 emiss=emiss, reflc=reflc, dvset=float(sc[0]), iout=iout, icp=icp, ibrd=ibrd,
                                 sclcpl=float(sc[1]), sclhw=float(sc[2]), y0res=float(sc[3]), cntnm=sc[5:12].copy()))
    return out


class Golden:
    def __init__(self, name: str, tmpdir: str):
        z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        self.name = name
        self.tape3 = os.path.join(tmpdir, f"TAPE3_{name}")
        with open(self.tape3, "wb") as f:
            f.write(z["tape3"].tobytes())
        self.profiles = read_case_bytes(z["case"].tobytes())
        xs = [k for k in z.files if k.startswith("xsfile_")]
        if xs:   # the synthetic cross-section library of the fixture (FSCDXS + xs files)
            self.xs_dir = os.path.join(tmpdir, f"xs_{name}")
            os.makedirs(self.xs_dir, exist_ok=True)
            for k in xs:
                with open(os.path.join(self.xs_dir, k[len("xsfile_"):]), "wb") as f:
                    f.write(z[k].tobytes())
            for pr in self.profiles:
                if pr.xs_names:
                    pr.xs_dir = self.xs_dir
        self.expected = []
        for i in range(int(z["nprof"])):
            kw = {k: z[f"p{i}_{k}"] for k in FIELDS}
            odx = z[f"p{i}_odxsec"] if f"p{i}_odxsec" in z.files else None
            self.expected.append(caseio.Dump(**kw, tmpsfc_out=float(z[f"p{i}_tmpsfc_out"]), odxsec=odx))


def max_rel(a: np.ndarray, b: np.ndarray, floor: float) -> float:
    """max |a-b| / max(|b|, floor): relative error with an absolute floor for values that
    underflow the physics (e.g. a 1e-250 transmittance)."""
    den = np.maximum(np.abs(b), floor)
    return float(np.max(np.abs(a - b) / den)) if a.size else 0.0


def compare(got: caseio.Dump, exp: caseio.Dump, rtol: float = RTOL, what: str = "", skip=(), rad_floor: float = 0.0):
    """Layer optical depths are compared relative to the total optical depth scale of the
    (layer, wavenumber) cell: a per-molecule term that is 1e-30 of the total cannot change any
    observable at 1e-6.  Spectral outputs use plain relative error (rad_floor: absolute floor for radiances that a
    REAL*4 output array cannot represent, e.g. a 1e-100 Planck radiance in the ultraviolet)."""
    errs = {}
    od_floor = 1e-12 * max(float(np.max(np.abs(exp.o))), 1e-300)
    errs["o"] = max_rel(got.o, exp.o, od_floor)
    tot = np.maximum(np.abs(exp.o), od_floor)[:, None, :]
    if "o_by_mol" not in skip:
        errs["o_by_mol"] = float(np.max(np.abs(got.o_by_mol - exp.o_by_mol) / np.maximum(np.abs(exp.o_by_mol), 1e-6 * tot)))
    errs["oc"] = float(np.max(np.abs(got.oc - exp.oc) / np.maximum(np.abs(exp.oc), 1e-6 * tot)))
    errs["o_clw"] = max_rel(got.o_clw, exp.o_clw, od_floor)
    if exp.odxsec is not None:
        assert got.odxsec is not None, what + ": no ODXSEC returned"
        errs["odxsec"] = max_rel(got.odxsec, exp.odxsec, od_floor)
    for k in ("rup", "rdn", "rad"):
        errs[k] = max_rel(getattr(got, k), getattr(exp, k), max(rad_floor, 1e-12 * max(float(np.max(np.abs(exp.rad))), 1e-300)))
    errs["trtot"] = max_rel(got.trtot, exp.trtot, 1e-12)
    errs["tb"] = max_rel(got.tb, exp.tb, 1e-3)
    errs["tmr"] = max_rel(got.tmr, exp.tmr, 1e-3)
    assert abs(got.tmpsfc_out - exp.tmpsfc_out) <= 1e-12 * max(1.0, abs(exp.tmpsfc_out)), what
    bad = {k: v for k, v in errs.items() if not (v <= rtol)}
    assert not bad, f"{what}: relative errors above {rtol:g}: {bad} (all: {errs})"
    return errs


def compare_nan_aware(got: caseio.Dump, exp: caseio.Dump, rtol: float = RTOL, what: str = "", rad_floor: float = 0.0):
    """For inputs on which the REFERENCE returns NaN somewhere (a NaN column amount: src/modm.f90:384,432 add the NaN term):
    the NaN positions of every field must coincide exactly, and the values elsewhere agree as compare() demands."""
    import dataclasses

    masked_g, masked_e = {}, {}
    for k in FIELDS:
        g, e = np.asarray(getattr(got, k), np.float64), np.asarray(getattr(exp, k), np.float64)
        ng, ne = np.isnan(g), np.isnan(e)
        assert np.array_equal(ng, ne), f"{what}: NaN pattern of {k} differs: {int(ng.sum())} here, {int(ne.sum())} in the reference"
        masked_g[k], masked_e[k] = np.where(ne, 0.0, g), np.where(ne, 0.0, e)
    assert any(np.isnan(np.asarray(getattr(exp, k), np.float64)).any() for k in FIELDS), what + ": the fixture holds no NaN"
    return compare(dataclasses.replace(got, **masked_g), dataclasses.replace(exp, **masked_e), rtol=rtol, what=what, rad_floor=rad_floor)


def per_molecule_errors(got: caseio.Dump, exp: caseio.Dump) -> np.ndarray:
    """[nmol] max relative error of O_BY_MOL per molecule over the cells where that molecule's optical depth is at least
    1e-6 of its own peak (NaN for a molecule without optical depth): a wrong partition sum, mass or width of ONE species shows
    here even when the species is a trace in the total."""
    nmol = exp.o_by_mol.shape[1]
    out = np.full(nmol, np.nan)
    for m in range(nmol):
        e, g = exp.o_by_mol[:, m, :], got.o_by_mol[:, m, :]
        pk = np.abs(e).max()
        if pk > 0:
            sel = np.abs(e) >= 1e-6 * pk
            out[m] = float(np.max(np.abs(g[sel] - e[sel]) / np.abs(e[sel])))
    return out
