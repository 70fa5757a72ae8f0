"""Seeded synthetic inputs for the MODM/RTM hot path.

The reference tree ships no line file (its TAPE3 is a dangling symlink) and no expected
outputs, so every parity and benchmark case is built from these generators.  The layer
quantities follow the reference's own conventions: pressures in mbar, temperatures in K,
column amounts in molecules/cm^2 (``src/monortm.f90:380-488``), layers ordered surface ->
top (``IDU=1``, ``src/RTMmono.f90:173``).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from .tape3 import LineRecords

RADCN2 = 1.4387752  # src/PhysConstants.f90:39


@dataclass
class Profile:
    """One atmospheric profile + viewing set-up, as PROGRAM MONORTM hands it to
    MODM / CALCTMR / RTM (``src/monortm.f90:557-574``)."""

    wn: np.ndarray          # [nwn] cm-1 ascending
    p: np.ndarray           # [nlay] mbar
    t: np.ndarray           # [nlay] K (layer mean)
    tz: np.ndarray          # [nlay+1] K (levels, 0 = surface)
    wkl: np.ndarray         # [nlay, nmol] molecules/cm^2
    wbrodl: np.ndarray      # [nlay]
    clw: np.ndarray         # [nlay] mm
    irt: int = 3            # 1 up, 2 limb, 3 down
    tmpsfc: float = 2.75
    emiss: np.ndarray | None = None
    reflc: np.ndarray | None = None
    dvset: float = 0.0
    iout: int = 1
    icp: int = 1
    ibrd: int = 0
    sclcpl: float = 1.0
    sclhw: float = 1.0
    y0res: float = 0.0
    cntnm: np.ndarray = field(default_factory=lambda: np.ones(7))
    # cross-section molecules (IXSECT = 1): names as on record 2.2.x and column amounts [nlay, nxs] (COMMON /PATHX/ XAMNT,
    # src/monortm.f90:492-530); the FSCDXS / xs files are looked up in `xs_dir`
    xs_names: list | None = None
    xamnt: np.ndarray | None = None
    xs_dir: str | None = None

    def __post_init__(self):
        self.wn = np.ascontiguousarray(self.wn, np.float64)
        nwn = len(self.wn)
        if self.emiss is None:
            self.emiss = np.ones(nwn)
        if self.reflc is None:
            self.reflc = np.zeros(nwn)
        for k in ("p", "t", "tz", "wkl", "wbrodl", "clw", "emiss", "reflc", "cntnm"):
            setattr(self, k, np.ascontiguousarray(getattr(self, k), np.float64))
        if self.xs_names is not None:
            self.xamnt = np.ascontiguousarray(self.xamnt, np.float64).reshape(len(self.p), len(self.xs_names))

    @property
    def ixsect(self):
        return 1 if self.xs_names else 0

    @property
    def nwn(self):
        return len(self.wn)

    @property
    def nlay(self):
        return len(self.p)

    @property
    def nmol(self):
        return self.wkl.shape[1]


def standard_atmosphere(nlay: int = 64, ztop_km: float = 32.0, nmol: int = 7):
    """Hydrostatic exponential atmosphere of SURVEY.md 8(d) c2: p=1013 exp(-z/7.5),
    T=max(288.2-6.5z, 216.7), H2O scale height 2 km, fixed dry mixing ratios."""
    zlev = np.linspace(0.0, ztop_km, nlay + 1)
    plev = 1013.0 * np.exp(-zlev / 7.5)
    tlev = np.maximum(288.2 - 6.5 * zlev, 216.7)
    zmid = 0.5 * (zlev[:-1] + zlev[1:])
    p = 0.5 * (plev[:-1] + plev[1:])
    t = 0.5 * (tlev[:-1] + tlev[1:])
    air = 2.1e25 * (plev[:-1] - plev[1:]) / 1013.0  # total column per layer
    h2o_vmr = 0.012 * np.exp(-zmid / 2.0) + 4e-6
    vmr = np.zeros((nlay, max(nmol, 7)))
    vmr[:, 0] = h2o_vmr
    vmr[:, 1] = 4e-4
    vmr[:, 2] = 3e-7 * (1 + zmid / 5.0)
    vmr[:, 3] = 3.2e-7
    vmr[:, 4] = 1.5e-7
    vmr[:, 5] = 1.7e-6
    vmr[:, 6] = 0.209
    wkl = vmr[:, :nmol] * air[:, None]
    wbrodl = 0.781 * air
    return dict(p=p, t=t, tz=tlev, wkl=wkl, wbrodl=wbrodl, clw=np.zeros(nlay), zmid=zmid)


def c2_channels(nchan: int = 50, seed: int = 20261003, lo: float = 0.3, hi: float = 30.0):
    rng = np.random.default_rng(seed + 17)
    return np.sort(rng.uniform(lo, hi, nchan))


def c2_profile(nchan: int = 50, seed: int = 20261003, nlay: int = 64) -> Profile:
    a = standard_atmosphere(nlay)
    return Profile(wn=c2_channels(nchan, seed), p=a["p"], t=a["t"], tz=a["tz"], wkl=a["wkl"],
                   wbrodl=a["wbrodl"], clw=a["clw"], irt=3)


def perturbed_profile(ip: int, wn: np.ndarray, nlay: int = 64, cloud: bool = False,
                      irt: int = 3, ztop_km: float = 32.0) -> Profile:
    """c4/c5 sonde-like member ``ip``: smooth T perturbation N(0,3K), H2O x lognormal(0.3),
    surface pressure x U(0.97,1.03); optional liquid cloud in 2-4 layers with 255<T<285."""
    rng = np.random.default_rng(1000 + ip)
    a = standard_atmosphere(nlay, ztop_km=ztop_km)
    knots = rng.normal(0.0, 3.0, 6)
    xk = np.linspace(0, nlay, 6)
    dT_lev = np.interp(np.arange(nlay + 1), xk, knots)
    tz = a["tz"] + dT_lev
    t = a["t"] + 0.5 * (dT_lev[:-1] + dT_lev[1:])
    ps = rng.uniform(0.97, 1.03)
    wkl = a["wkl"] * ps
    wkl[:, 0] *= rng.lognormal(0.0, 0.3)
    clw = np.zeros(nlay)
    if cloud:
        ok = np.flatnonzero((t > 255.0) & (t < 285.0))
        k = int(rng.integers(2, 5))
        sel = rng.choice(ok, size=min(k, len(ok)), replace=False)
        clw[sel] = rng.uniform(0.0, 0.05, len(sel))
    kw = {}
    if irt == 1:
        kw = dict(tmpsfc=290.0, emiss=np.full(len(wn), 0.6), reflc=np.full(len(wn), 0.4))
    return Profile(wn=wn, p=a["p"] * ps, t=t, tz=tz, wkl=wkl, wbrodl=a["wbrodl"] * ps, clw=clw,
                   irt=irt, **kw)


def synthetic_lines(n: int = 500, seed: int = 20261003, vlo: float = 0.05, vhi: float = 54.9,
                    sdep_frac: float = 0.0, lc_frac: float = 0.0) -> LineRecords:
    """Random line list of SURVEY.md 8(d): molecule mix {H2O x3, O3 x2, O2, N2O, CO2}/8,
    isotope 1.  ``sp`` is stored the way LNFL stores it: HITRAN S divided by
    nu (1 - exp(-c2 nu/296)) (``src/modm.f90:372`` applies the inverse).

    ``lc_frac`` of the O2 lines get first-order line-coupling records (IFLG=1 followed by
    an IFLG=-1 record holding Y,G at 200/250/296/340 K); ``sdep_frac`` of all lines get a
    speed-dependence parameter."""
    rng = np.random.default_rng(seed)
    vnu = np.sort(rng.uniform(vlo, vhi, n))
    u_s = rng.uniform(0.0, 1.0, n)
    alfa = rng.uniform(0.03, 0.11, n)
    hwhm = rng.uniform(0.03, 0.5, n)
    epp = rng.uniform(0.0, 2000.0, n)
    tmpalf = rng.uniform(0.4, 0.8, n)
    pshift = rng.uniform(-0.003, 0.003, n)
    molmix = np.array([1, 1, 1, 3, 3, 7, 4, 2])
    mol = molmix[rng.integers(0, len(molmix), n)]
    # O2 air width must exceed 0.21*self so the foreign width stays positive
    # (src/lnfl_mod.f90:98-101)
    o2 = mol == 7
    hwhm[o2] = rng.uniform(0.03, 0.06, int(o2.sum()))
    alfa[o2] = rng.uniform(0.04, 0.06, int(o2.sum()))
    # log10 strength range per molecule, sized to the column amounts of standard_atmosphere()
    # so that zenith optical depths land in ~1e-3 .. 3 (a fully opaque atmosphere would make
    # brightness temperature insensitive to optical-depth errors)
    srange = {1: (-28.0, -24.3), 2: (-28.0, -24.5), 3: (-24.0, -20.5), 4: (-24.0, -20.5),
              7: (-30.0, -26.5)}
    lo_s = np.array([srange[int(m)][0] for m in mol])
    hi_s = np.array([srange[int(m)][1] for m in mol])
    s_hitran = 10.0 ** (lo_s + (hi_s - lo_s) * u_s)
    sp = s_hitran / (vnu * (1.0 - np.exp(-RADCN2 * vnu / 296.0)))
    sdep = np.zeros(n)
    if sdep_frac > 0:
        pick = rng.random(n) < sdep_frac
        sdep[pick] = rng.uniform(0.05, 0.15, int(pick.sum()))
    iflg = np.zeros(n, np.int32)
    lcsel = np.zeros(n, bool)
    if lc_frac > 0:
        lcsel = o2 & (rng.random(n) < lc_frac)
        iflg[lcsel] = 1
    cols = dict(vnu=[], sp=[], alfa=[], epp=[], mol=[], hwhm=[], tmpalf=[], pshift=[], iflg=[],
                sdep=[])
    for i in range(n):
        cols["vnu"].append(vnu[i]); cols["sp"].append(sp[i]); cols["alfa"].append(alfa[i])
        cols["epp"].append(epp[i]); cols["mol"].append(int(mol[i]) + 100)
        cols["hwhm"].append(hwhm[i]); cols["tmpalf"].append(tmpalf[i])
        cols["pshift"].append(pshift[i]); cols["iflg"].append(int(iflg[i]))
        cols["sdep"].append(sdep[i])
        if lcsel[i]:
            # small, smoothly T-dependent Y (1/atm-like) and G; physically sized so the
            # coupled O2 band stays positive
            y = rng.uniform(-0.3, 0.3) * np.array([1.3, 1.15, 1.0, 0.9])
            g = rng.uniform(-0.02, 0.02) * np.array([1.5, 1.2, 1.0, 0.8])
            cols["vnu"].append(y[0]); cols["sp"].append(g[0]); cols["alfa"].append(y[1])
            cols["epp"].append(g[1])
            cols["mol"].append(int(np.float32(y[2]).view(np.int32)))
            cols["hwhm"].append(g[2]); cols["tmpalf"].append(y[3]); cols["pshift"].append(g[3])
            cols["iflg"].append(-1); cols["sdep"].append(0.0)
    return LineRecords(**{k: np.asarray(v) for k, v in cols.items()})


# isotopologues per molecule known to TIPS_2003 (src/tips_2003.f90:361-369)
ISONM = (6, 9, 18, 5, 6, 3, 3, 3, 2, 1, 2, 1, 3, 1, 2, 2, 1, 2, 5, 3, 2, 1, 3, 2, 1, 2, 1, 1, 1, 1, 3, 1, 1, 1, 2, 1, 2, 2, 1)


def trace_columns(a: dict, nmol: int, seed: int = 0) -> np.ndarray:
    """Column amounts [nlay, nmol] for nmol > 7: the seven majors of standard_atmosphere() plus N2 (molecule 22) and trace
    species at 1e-9 .. 1e-7 of the air column."""
    rng = np.random.default_rng(seed + 77)
    nlay = len(a["p"])
    air = a["wbrodl"] / 0.781
    wkl = np.zeros((nlay, nmol))
    wkl[:, :7] = a["wkl"][:, :7]
    for m in range(8, nmol + 1):
        wkl[:, m - 1] = air * 10 ** rng.uniform(-9.0, -7.0) * (1 + 0.3 * np.sin(np.arange(nlay) / (2.0 + m % 5)))
    if nmol >= 22:
        wkl[:, 21] = 0.781 * air
    return wkl


def all_molecule_lines(n: int, seed: int, nmol: int = 39, vlo: float = 0.05, vhi: float = 54.9, skip=(19, 20),
                       col: np.ndarray | None = None, sdep_frac: float = 0.1) -> LineRecords:
    """Random list with lines of EVERY molecule 1..nmol (except `skip`) and isotopologues up to min(9, ISONM).  Molecules > 7
    get hwhm = alfa: the reference indexes its 7-element rho_molec with the molecule number (src/modm.f90:845), and with equal
    self and foreign widths the out-of-bounds value multiplies zero (the reference returns NaN for molecules 19 and 20 all the
    same).  Strengths are sized to the surface-layer column `col[nmol]` so that every species reaches a layer optical depth
    of 1e-3 .. 1e-1."""
    rng = np.random.default_rng(seed)
    mols = np.array([m for m in range(1, nmol + 1) if m not in skip])
    mol = np.concatenate([mols, rng.choice(mols, max(0, n - len(mols)))])[:max(n, 1)]
    rng.shuffle(mol)
    n = len(mol)
    iso = np.array([int(rng.integers(1, min(9, ISONM[m - 1]) + 1)) for m in mol])
    vnu = np.sort(rng.uniform(vlo, vhi, n))
    alfa = rng.uniform(0.04, 0.11, n)
    hwhm = np.where(mol <= 6, rng.uniform(0.05, 0.45, n), alfa)
    o2 = mol == 7
    alfa[o2] = rng.uniform(0.04, 0.06, int(o2.sum()))
    hwhm[o2] = rng.uniform(0.03, 0.06, int(o2.sum()))
    n2 = mol == 22          # air width -> foreign width with rvmr = 0.79 (src/lnfl_mod.f90:102-113): keep it positive
    hwhm[n2] = alfa[n2]
    if col is None:
        col = np.full(nmol, 1e17)
    s_hitran = 10 ** rng.uniform(-3.0, -1.0, n) * np.pi * 0.08 / np.maximum(col[mol - 1], 1e-30)
    sp = s_hitran / (vnu * (1.0 - np.exp(-RADCN2 * vnu / 296.0)))
    sdep = np.where(rng.random(n) < sdep_frac, rng.uniform(0.05, 0.15, n), 0.0)
    return LineRecords(vnu=vnu, sp=sp, alfa=alfa, epp=rng.uniform(0.0, 1500.0, n), mol=mol + 100 * iso, hwhm=hwhm,
                       tmpalf=rng.uniform(0.4, 0.8, n), pshift=rng.uniform(-0.003, 0.003, n), iflg=np.zeros(n, np.int32), sdep=sdep)
