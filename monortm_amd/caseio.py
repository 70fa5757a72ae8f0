"""Binary case / dump files exchanged with the Fortran dump harness (examples/harness.f90).

Plain data formats only (little-endian stream, documented in the harness header); used by
the tests, by fixture generation and by bench.py's CPU-baseline leg.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass

import numpy as np

from .synth import Profile

MAGIC = 1297241155  # b'CTRM'


def write_case(path: str, profiles: list[Profile]) -> None:
    with open(path, "wb") as f:
        f.write(struct.pack("<ii", MAGIC, len(profiles)))
        for pr in profiles:
            f.write(struct.pack("<8i", pr.nwn, pr.nlay, pr.nmol, pr.irt, pr.iout, pr.icp,
                                pr.ibrd, pr.ixsect))
            sc = np.array([pr.dvset, pr.sclcpl, pr.sclhw, pr.y0res, pr.tmpsfc, *pr.cntnm],
                          np.float64)
            assert sc.size == 12
            f.write(sc.tobytes())
            for a in (pr.wn, pr.p, pr.t, pr.clw, pr.wbrodl, pr.tz, pr.wkl, pr.emiss, pr.reflc):
                f.write(np.ascontiguousarray(a, np.float64).tobytes())
            if pr.ixsect:   # cross-section molecules: names (10 characters each) and amounts [nlay][nxs]
                f.write(struct.pack("<i", len(pr.xs_names)))
                for n in pr.xs_names:
                    f.write(n.encode("ascii")[:10].ljust(10))
                f.write(np.ascontiguousarray(pr.xamnt, np.float64).tobytes())


@dataclass
class Dump:
    """Outputs of one profile; arrays are C-ordered with the wavenumber axis last."""

    o: np.ndarray         # [nlay, nwn]
    o_by_mol: np.ndarray  # [nlay, nmol, nwn]
    oc: np.ndarray        # [nlay, 5, nwn]  continuum of molecules 1,2,3,7,22
    o_clw: np.ndarray     # [nlay, nwn]
    rup: np.ndarray
    rdn: np.ndarray
    trtot: np.ndarray
    rad: np.ndarray
    tb: np.ndarray
    tmr: np.ndarray
    tmpsfc_out: float
    odxsec: np.ndarray | None = None   # [nlay, nwn] total optical depth of the cross-section molecules (IXSECT = 1 only)


def read_dump(path: str) -> list[Dump]:
    data = open(path, "rb").read()
    pos = 0
    out = []
    while pos < len(data):
        nwn, nlay, nmol = struct.unpack_from("<3i", data, pos)
        pos += 12

        def take(*shape):
            nonlocal pos
            n = int(np.prod(shape))
            a = np.frombuffer(data, np.float64, n, pos).reshape(shape).copy()
            pos += 8 * n
            return a

        o = take(nlay, nwn)
        obm = take(nlay, nmol, nwn)
        oc = take(nlay, 5, nwn)
        oclw = take(nlay, nwn)
        rup, rdn, trtot, rad, tb, tmr = (take(nwn) for _ in range(6))
        (ts,) = take(1)
        odx = None
        if pos < len(data) and struct.unpack_from("<i", data, pos)[0] == -7777:   # marker of the optional ODXSEC block
            pos += 4
            odx = take(nlay, nwn)
        out.append(Dump(o, obm, oc, oclw, rup, rdn, trtot, rad, tb, tmr, float(ts), odx))
    return out


def write_dump(path: str, dumps: list[Dump]) -> None:
    with open(path, "wb") as f:
        for d in dumps:
            nlay, nmol, nwn = d.o_by_mol.shape
            f.write(struct.pack("<3i", nwn, nlay, nmol))
            for a in (d.o, d.o_by_mol, d.oc, d.o_clw, d.rup, d.rdn, d.trtot, d.rad, d.tb, d.tmr):
                f.write(np.ascontiguousarray(a, np.float64).tobytes())
            f.write(struct.pack("<d", d.tmpsfc_out))
            if d.odxsec is not None:
                f.write(struct.pack("<i", -7777))
                f.write(np.ascontiguousarray(d.odxsec, np.float64).tobytes())
