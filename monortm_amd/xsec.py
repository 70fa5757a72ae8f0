"""Cross-section ("xs") molecule data: the FSCDXS master file and the per-temperature xs files as the reference reads them,
a writer for synthetic ones (the reference tree ships neither), and the table loader that feeds the C ABI.

Formats (reference src/monortm_sub.F90, paths relative to /root/reference):
  * FSCDXS (XSREAD, :1339-1396): two header lines (``READ (IXFIL,905)`` with ``905 FORMAT (/)``), then one record per
    (molecule, spectral region) in format 915 ``(A10,2F10.4,F10.8,I5,5X,I5,A1,4X,6A10)``: name, V1, V2, DV, number of
    temperatures, format code, format letter, up to six file names in ASCENDING temperature.  A line starting with ``*`` is a
    comment, one starting with ``%`` ends the file.
  * xs file (MONORTM_XSEC_SUB, :1660-1672): header in format 910 ``(A10,2F10.4,I10,3G10.3,3A10)``: molecule, V1, V2, number of
    points, temperature [K], pressure, maximum, three 10-character source words - the third one ``      TORR`` means the
    pressure is in torr, anything else millibar - followed by the values, list directed.
  * the molecule names and column amounts come with the profile (records 2.2.x of MONORTM_PROF.IN, src/monortm.f90:492-530).

Aliases and molecular masses: BLOCK DATA BXSECT (:1424-1483).  A spectral region is kept when V2 > min(wn) and V1 < max(wn)
(:1364); at most six regions per molecule.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field

import numpy as np

# (alias 1..4), mass - BLOCK DATA BXSECT, src/monortm_sub.F90:1441-1472
XS_SPECIES = [
    (("CLONO2", "CLNO3", "", ""), 97.46), (("HNO4", "", "", ""), 79.01), (("CHCL2F", "CFC21", "CFC21", "F21"), 102.92),
    (("CCL4", "", "", ""), 153.82), (("CCL3F", "CFCL3", "CFC11", "F11"), 137.37), (("CCL2F2", "CF2CL2", "CFC12", "F12"), 120.91),
    (("C2CL2F4", "C2F4CL2", "CFC114", "F114"), 170.92), (("C2CL3F3", "C2F3CL3", "CFC113", "F113"), 187.38),
    (("N2O5", "", "", ""), 108.01), (("HNO3", "", "", ""), 63.01), (("CF4", "", "CFC14", "F14"), 88.00),
    (("CHCLF2", "CHF2CL", "CFC22", "F22"), 86.47), (("CCLF3", "", "CFC13", "F13"), 104.46), (("C2CLF5", "", "CFC115", "F115"), 154.47),
    (("NO2", "", "", ""), 45.99),
]


def species_index(name: str) -> int:
    """0-based index into XS_SPECIES of a (left-justified, upper-case) name; KeyError like the reference's STOP."""
    n = name.strip()
    for i, (al, _) in enumerate(XS_SPECIES):
        if n and n in al:
            return i
    raise KeyError(f"{name!r} is not one of the cross-section molecules (XSREAD: STOPPED IN XSREAD)")


@dataclass
class XsRegion:
    """One spectral region of one molecule as the reference holds it after reading the files."""
    v1: float              # V1FX, V2FX of the FSCDXS entry: the +- 1 cm-1 test that decides whether the region is processed
    v2: float              # (src/monortm_sub.F90:1645)
    temps: np.ndarray      # [ntemp] K, ascending
    pres_mb: np.ndarray    # [ntemp] measurement pressures in millibar
    data: list             # [ntemp] spectra, each with the number of points its own file header states
    xdoplr: float          # Doppler half width at 296 K at the region centre (XSREAD :1386)
    v1h: float = 0.0       # V1, V2, NPTS on the header of the LAST file read: the grid of every spectrum of the region and the
    v2h: float = 0.0       # in-range test of a wavenumber (:1663-1666, :1709, :1789)
    npts: int = 0


@dataclass
class XsTables:
    names: list[str]
    regions: list[list[XsRegion]] = field(default_factory=list)   # per molecule

    def flatten(self):
        """Arrays for the C ABI: region table [nreg, 8] = (molecule, V1FX, V2FX, npts, ntemp, xdoplr, V1 and V2 of the last
        file's header), temps / pressures [nreg, 6], offsets [nreg, 6] into the value pool.  Every spectrum occupies `npts` values
        of the pool (the last file's count): a shorter file is padded with zeros, as the reference's work array would hold zeros
        (or stale values) beyond what that file filled."""
        reg, temps, pres, offs, pool = [], [], [], [], []
        pos = 0
        for m, rs in enumerate(self.regions):
            for r in rs:
                nt, npts = len(r.data), (r.npts or len(r.data[-1]))
                reg.append((m, r.v1, r.v2, npts, nt, r.xdoplr, r.v1h if r.npts else r.v1, r.v2h if r.npts else r.v2))
                t = np.zeros(6)
                p = np.zeros(6)
                o = np.zeros(6, np.int64)
                t[:nt], p[:nt] = r.temps, r.pres_mb
                for k in range(nt):
                    o[k] = pos
                    d = np.zeros(npts)
                    d[:min(npts, len(r.data[k]))] = np.asarray(r.data[k], np.float64)[:npts]
                    pool.append(d)
                    pos += npts
                temps.append(t)
                pres.append(p)
                offs.append(o)
        n = len(reg)
        return (np.array(reg, np.float64).reshape(n, 8), np.array(temps).reshape(n, 6), np.array(pres).reshape(n, 6),
                np.array(offs, np.int64).reshape(n, 6), np.concatenate(pool) if pool else np.zeros(0))


# ------------------------------------------------------------------------------------------------------------------
# writers (synthetic data)
# ------------------------------------------------------------------------------------------------------------------
def write_fscdxs(path: str, entries: list[tuple]) -> None:
    """entries: (name, v1, v2, dv, [file names in ascending temperature]) - format 915 of XSREAD."""
    with open(path, "w") as f:
        f.write("  synthetic cross-section master file (format of LBLRTM's FSCDXS)\n")
        f.write("  NAME          V1        V2        DV  NT      FRM      FILES\n")
        for name, v1, v2, dv, files in entries:
            assert len(files) <= 6
            f.write(f"{name:<10s}{v1:10.4f}{v2:10.4f}{dv:10.8f}{len(files):5d}     {91:5d}{'N':1s}    " + "".join(f"{x:<10s}" for x in files) + "\n")
        f.write("%\n")


def write_xs_file(path: str, name: str, v1: float, v2: float, temp: float, pres: float, values: np.ndarray, torr: bool = True) -> None:
    """format 910 header + list-directed values."""
    values = np.asarray(values, np.float64)
    src = ("  SYNTHETIC", "          ", "      TORR" if torr else "        MB")
    with open(path, "w") as f:
        f.write(f"{name:<10s}{v1:10.4f}{v2:10.4f}{len(values):10d}{temp:10.3f}{pres:10.3f}{values.max():10.3E}" + "".join(src) + "\n")
        for i in range(0, len(values), 8):
            f.write(" ".join(f"{x:.9E}" for x in values[i:i + 8]) + "\n")


# ------------------------------------------------------------------------------------------------------------------
# reader (what XSREAD + the file loop of MONORTM_XSEC_SUB leave in memory)
# ------------------------------------------------------------------------------------------------------------------
def _fields(line: str, widths):
    out, pos = [], 0
    for w in widths:
        out.append(line[pos:pos + w])
        pos += w
    return out


def load_tables(directory: str, names: list[str], wn_min: float, wn_max: float) -> XsTables:
    """Parse FSCDXS and the xs files of the regions that overlap [wn_min, wn_max] for the requested molecules."""
    idx = [species_index(n) for n in names]
    tabs = XsTables(names=[n.strip() for n in names], regions=[[] for _ in names])
    lines = open(os.path.join(directory, "FSCDXS")).read().split("\n")[2:]
    found = [False] * len(names)
    for ln in lines:
        if ln.startswith("*"):
            continue
        if ln.startswith("%") or not ln.strip():
            break
        ln = ln.ljust(120)
        f = _fields(ln, (10, 10, 10, 10, 5, 5, 5, 1, 4, 10, 10, 10, 10, 10, 10))
        xname, v1x, v2x, ntemp = f[0].strip(), float(f[1]), float(f[2]), int(f[4])
        files = [x.strip() for x in f[9:9 + ntemp]]
        for i, k in enumerate(idx):
            if xname and xname in XS_SPECIES[k][0]:
                found[i] = True
                if v2x > wn_min and v1x < wn_max:
                    if len(tabs.regions[i]) >= 5:   # (the tables of COMMON /XSECTR/ hold five regions per molecule)
                        raise ValueError("XSREAD - NSPECR .GT. 5")
                    temps, pres, data = [], [], []
                    v1h = v2h = 0.0
                    npts = 0
                    for fn in files:
                        body = open(os.path.join(directory, fn)).read().split("\n")
                        h = _fields(body[0].ljust(100), (10, 10, 10, 10, 10, 10, 10, 10, 10, 10))
                        v1h, v2h = float(h[1]), float(h[2])   # (the LAST file's header decides, :1663-1666)
                        npts, tx, pr = int(h[3]), float(h[4]), float(h[5])
                        vals = np.array(" ".join(body[1:]).split()[:npts], np.float64)
                        assert len(vals) == npts
                        temps.append(tx)
                        pres.append(pr * (1013. / 760) if h[9] == "      TORR" else pr)   # PTORMB, :1626
                        data.append(vals)
                    # 3.58115E-07 = SQRT(2 LOG(2) AVOGAD BOLTZ / CLIGHT**2), T296 = 296 (XSREAD :1383-1387)
                    xdop = 3.58115E-07 * (0.5 * (v1x + v2x)) * np.sqrt(296.0 / XS_SPECIES[k][1])
                    tabs.regions[i].append(XsRegion(v1x, v2x, np.array(temps), np.array(pres), data, float(xdop), v1h, v2h, npts))
    if not all(found):
        raise ValueError("molecule not found on FSCDXS (IXFLAG - XSREAD)")
    return tabs


def synthetic_library(directory: str, seed: int = 7, f12_pres_mb: float = 20.0, fscdxs_pad: tuple = (0.0, 0.0), ragged: int = 0) -> list[str]:
    """A small FSCDXS + xs files in the thermal infrared: CCL4 (one region, three temperatures), F11 (two regions, two
    temperatures each, one file with its pressure in millibar) and F12 (one region, one temperature, measured at
    `f12_pres_mb`).  Cross sections in cm^2/molecule, smooth band shapes with fine structure so that the pressure convolution
    matters.  Measurement pressures are low: the reference's convolve() resamples every spectrum on a grid of a quarter of the
    EXTRA Lorentz width into a 10^7-element array (src/monortm_sub.F90:1758,:1773-1786) and overruns it for layers whose
    pressure is below that of the measurement - fixtures must keep every layer above it.  fscdxs_pad = (below, above): the
    FSCDXS entries state bounds that much wider than the file headers (real master files carry rounded bounds: the FSCDXS pair
    decides whether a region is processed, the header pair is the grid).  ragged > 0: the FIRST temperature file of every
    region with several temperatures holds that many points fewer than the last one (the packers must place every spectrum at
    a stride of the last file's count, zero-filled).  Returns the names."""
    rng = np.random.default_rng(seed)
    os.makedirs(directory, exist_ok=True)

    def band(v, centre, width, peak):
        x = (v - centre) / width
        return peak * (np.exp(-x * x) * (1 + 0.3 * np.sin(37 * x) + 0.2 * np.cos(11 * x)) + 0.02)

    ent = []
    spec = [("CCL4", 770.0, 812.0, 0.02, [(208.0, 5.0, True), (253.0, 20.0, True), (297.0, 60.0, True)], 793.0, 6.0, 5e-18),
            ("F11", 830.0, 860.0, 0.015, [(216.0, 30.0, True), (296.0, 90.0, False)], 846.0, 4.0, 4e-18),
            ("F11", 1060.0, 1107.0, 0.025, [(216.0, 20.0, True), (296.0, 50.0, True)], 1085.0, 7.0, 2e-18),
            ("F12", 850.0, 950.0, 0.05, [(270.0, f12_pres_mb, False)], 921.0, 9.0, 3e-18)]
    for k, (name, v1, v2, dv, tps, c, w, pk) in enumerate(spec):
        npts = int(round((v2 - v1) / dv)) + 1
        v = v1 + dv * np.arange(npts)
        files = []
        for (tt, pp, torr) in tps:
            fn = f"xs{name.lower()}{k}t{int(tt)}"[:10]
            shape = band(v, c, w * (tt / 296.0) ** 0.5, pk * (296.0 / tt) ** 0.7) * (1 + 0.05 * rng.standard_normal(npts).cumsum() / np.sqrt(npts))
            vals = np.maximum(shape, pk * 1e-3)
            if ragged and len(tps) > 1 and (tt, pp, torr) == tps[0]:
                vals = vals[: npts - ragged]
            write_xs_file(os.path.join(directory, fn), name, v1, v1 + dv * (len(vals) - 1), tt, pp, vals, torr)
            files.append(fn)
        ent.append((name, v1 - fscdxs_pad[0], v1 + dv * (npts - 1) + fscdxs_pad[1], dv, files))
    write_fscdxs(os.path.join(directory, "FSCDXS"), ent)
    return ["CCL4", "F11", "F12"]
