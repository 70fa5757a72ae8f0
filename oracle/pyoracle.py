"""TEST INFRASTRUCTURE - ctypes front end of the C restatement (oracle/liboracle.so).

May be imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product (monortm_amd) never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_dp = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "monortm_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "oracle"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.orc_load_tape3.argtypes = [C.c_char_p, C.c_double, C.c_double, C.POINTER(C.c_void_p)]
        L.orc_load_tape3.restype = C.c_int
        L.orc_load_tape3_kind.argtypes = [C.c_char_p, C.c_double, C.c_double, C.c_int, C.POINTER(C.c_void_p)]
        L.orc_load_tape3_kind.restype = C.c_int
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_last_error.argtypes = [C.c_void_p]
        L.orc_last_error.restype = C.c_char_p
        L.orc_nlines.argtypes = [C.c_void_p, C.c_int]
        L.orc_nlines.restype = C.c_int
        L.orc_modm.argtypes = [C.c_void_p, C.c_int, _dp, C.c_double, C.c_int, _dp, _dp, _dp, C.c_int, _dp, _dp,
                               C.c_double, C.c_double, C.c_double, _dp, C.c_int, _dp, _dp, _dp, _dp]
        L.orc_modm.restype = C.c_int
        L.orc_calctmr.argtypes = [C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp]
        L.orc_calctmr.restype = None
        L.orc_rtm.argtypes = [C.c_int, C.c_int, C.c_int, _dp, C.c_int, _dp, _dp, _dp, C.POINTER(C.c_double),
                              _dp, _dp, _dp, _dp, _dp, _dp, _dp]
        L.orc_rtm.restype = C.c_int
        L.orc_stats.argtypes = [C.POINTER(C.c_longlong), C.c_int]
        L.orc_stats.restype = None
        _lp = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
        L.orc_xsec.argtypes = [C.c_int, _dp, C.c_int, _dp, _dp, C.c_int, C.c_int, _dp, _dp, _dp, _lp, _dp, _dp, _dp]
        L.orc_xsec.restype = C.c_int
        L.orc_set_odxsec.argtypes = [C.c_void_p]
        L.orc_set_odxsec.restype = None
        L.orc_iso_stats.argtypes = [C.POINTER(C.c_longlong), C.c_int]
        L.orc_iso_stats.restype = None
        L.orc_kat.argtypes = [C.c_int, C.c_int, _dp, _dp, _dp]
        L.orc_kat.restype = None
        L.orc_kat_wide.argtypes = [C.c_int, C.c_int, _dp, _dp]
        L.orc_kat_wide.restype = None
        _LIB = L
    return _LIB


class OracleError(RuntimeError):
    pass


class Oracle:
    """One loaded TAPE3 (the reference loads it once per process with the first call's
    v1,v2 - src/modm.f90:187-190)."""

    def __init__(self, tape3: str, v1: float, v2: float, real_kind: int = 8):
        """real_kind: which build of the reference READS the line file (the owner rule of a coupling record that is the first
        record of a block differs: oracle/monortm_oracle.c orc_load_tape3_kind); the arithmetic is double either way."""
        self.L = lib()
        self.ctx = C.c_void_p()
        rc = self.L.orc_load_tape3_kind(tape3.encode(), v1, v2, real_kind, C.byref(self.ctx))
        if rc:
            msg = self.L.orc_last_error(self.ctx).decode()
            raise OracleError(f"orc_load_tape3 rc={rc}: {msg}")

    def close(self):
        if self.ctx:
            self.L.orc_free(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def census(self, reset: bool = True) -> dict:
        """Branch counts of the LINES walk since the last reset (cut-pass fraction and the
        Lorentz / Voigt split): line visits, visits rejected by the 25 cm-1 test, Lorentz and Voigt shape calls."""
        buf = (C.c_longlong * 16)()
        self.L.orc_stats(buf, int(reset))
        v = list(buf)
        return {"visits": v[0], "cut_rejected": v[1], "lorentz": v[2], "voigt": v[3], "coupled": v[12],
                "w4_region": v[4:8], "sd_region": v[8:12], "coupled_m3": v[13], "coupled_m5": v[14], "coupled_voigt": v[15]}

    def iso_census(self, reset: bool = True) -> np.ndarray:
        """[39, 9] evaluated shapes per (molecule, isotopologue) since the last reset: which TIPS / mass slots were visited."""
        buf = (C.c_longlong * (39 * 9))()
        self.L.orc_iso_stats(buf, int(reset))
        return np.array(list(buf), np.int64).reshape(39, 9)

    def nlines(self, mol: int) -> int:
        return self.L.orc_nlines(self.ctx, mol)

    def run(self, pr):
        """pr: monortm_amd.synth.Profile -> monortm_amd.caseio.Dump (MODM + CALCTMR + RTM)."""
        from monortm_amd.caseio import Dump

        nwn, nlay, nmol = pr.nwn, pr.nlay, pr.nmol
        odx = None
        if getattr(pr, "xs_names", None):   # IXSECT = 1: MONORTM_XSEC_SUB first (modm.f90:197), its sum enters O (:268)
            from monortm_amd import xsec

            tabs = xsec.load_tables(pr.xs_dir, pr.xs_names, float(pr.wn.min()), float(pr.wn.max()))
            reg, temps, pres, offs, pool = tabs.flatten()
            odx = np.zeros((nlay, nwn))
            if len(pool) == 0:
                pool = np.zeros(1)
            self.L.orc_xsec(nwn, pr.wn, nlay, pr.p, pr.t, len(pr.xs_names), len(reg), np.ascontiguousarray(reg.reshape(-1, 8) if len(reg) else np.zeros((1, 8))),
                            np.ascontiguousarray(temps.reshape(-1, 6) if len(reg) else np.zeros((1, 6))),
                            np.ascontiguousarray(pres.reshape(-1, 6) if len(reg) else np.zeros((1, 6))),
                            np.ascontiguousarray(offs.reshape(-1, 6) if len(reg) else np.zeros((1, 6), np.int64)), pool,
                            np.ascontiguousarray(pr.xamnt), odx)
            self.L.orc_set_odxsec(odx.ctypes.data_as(C.c_void_p))
        o = np.zeros((nlay, nwn))
        obm = np.zeros((nlay, nmol, nwn))
        oc = np.zeros((nlay, 5, nwn))
        oclw = np.zeros((nlay, nwn))
        rc = self.L.orc_modm(self.ctx, nwn, pr.wn, pr.dvset, nlay, pr.p, pr.t, pr.clw, nmol,
                             np.ascontiguousarray(pr.wkl), pr.wbrodl, pr.sclcpl, pr.sclhw, pr.y0res,
                             np.ascontiguousarray(pr.cntnm), pr.ibrd, o, obm, oc, oclw)
        if rc:
            raise OracleError(f"orc_modm rc={rc}: {self.L.orc_last_error(self.ctx).decode()}")
        tmr = np.zeros(nwn)
        self.L.orc_calctmr(nlay, nwn, pr.wn, pr.t, pr.tz, o, tmr)
        rup, rdn, trtot, rad, tb = (np.zeros(nwn) for _ in range(5))
        ts = C.c_double(pr.tmpsfc)
        self.L.orc_rtm(pr.iout, pr.irt, nwn, pr.wn, nlay, pr.t, pr.tz, o, C.byref(ts), rup, trtot, rdn,
                       pr.reflc, pr.emiss, rad, tb)
        return Dump(o, obm, oc, oclw, rup, rdn, trtot, rad, tb, tmr, ts.value, odx)


def kat_wide(which: int, args: np.ndarray) -> np.ndarray:
    """Known answers of the functions with long argument lists (INTENS, HALFWHM_C, LSF_LORTZ, LSF_SDVOIGT): args [n, 12] -> [n]
    (see orc_kat_wide in monortm_oracle.c)."""
    a = np.ascontiguousarray(args, np.float64)
    assert a.ndim == 2 and a.shape[1] == 12
    out = np.zeros(len(a))
    lib().orc_kat_wide(which, len(a), a, out)
    return out


def kat(which: int, args: np.ndarray, tab: np.ndarray | None = None) -> np.ndarray:
    """Function-level known answers of the C restatement: args [n, 4] -> [n, 2] (see orc_kat in monortm_oracle.c)."""
    a = np.ascontiguousarray(args, np.float64)
    out = np.zeros((len(a), 2))
    lib().orc_kat(which, len(a), a, np.ascontiguousarray(tab if tab is not None else np.zeros(119), np.float64), out)
    return out
