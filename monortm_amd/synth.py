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


# ----------------------------------------------------------------------------------------------------------------------
# A line list shaped like a real aer_v_3.x 0-55 cm-1 file (the real one is not in the tree): clustered, several
# isotopologues, the O2 60-GHz complex with first-order coupling records, lines of molecules beyond NMOL = 7.
# Line positions are the published rest frequencies (GHz, rounded) of the strong microwave lines - physical constants,
# e.g. Liebe et al. 1992 for O2 and H2O; strengths, widths and coupling coefficients are SYNTHETIC but sized like the real
# ones, so that optical depths and brightness temperatures are realistic in scale.
# ----------------------------------------------------------------------------------------------------------------------
GHZ_PER_WN = 29.9792458
O2_NPLUS_GHZ = (56.2648, 58.4466, 59.5910, 60.4348, 61.1506, 61.8002, 62.4112, 62.9980, 63.5685, 64.1278, 64.6789, 65.2241,
                65.7648, 66.3021, 66.8368, 67.3696, 67.9009, 68.4310, 68.9603, 69.4891)
O2_NMINUS_GHZ = (118.7503, 62.4863, 60.3061, 59.1642, 58.3239, 57.6125, 56.9682, 56.3634, 55.7838, 55.2214, 54.6712, 54.1300,
                 53.5957, 53.0669, 52.5424, 52.0214, 51.5034, 50.9877, 50.4742, 49.9618)
O2_SUBMM_GHZ = (368.4984, 424.7631, 487.2494, 715.3931, 773.8397, 834.1453, 1120.715, 1406.37)
H2O_GHZ = (22.2351, 183.3101, 321.2256, 325.1529, 380.1974, 439.1508, 443.0183, 448.0011, 470.8890, 474.6891, 488.4911,
           556.9360, 620.7008, 752.0332, 916.1716, 970.3150, 987.9268, 1097.3648, 1113.3430, 1153.1268, 1162.9116, 1207.6387,
           1228.7888, 1410.6180, 1602.2194)


def realistic_lines(seed: int = 5005, n_o3: int = 4300, n_trace: int = 500, ibrd_frac: float = 0.05) -> tuple[LineRecords, dict]:
    """~5400 records (>= 21 blocks of 250): O2 60-GHz complex with IFLG = 1 / -1 coupling pairs + the non-resonant term
    (IFLG = 3 / -3), O2 118 GHz and sub-millimetre lines, 16O18O lines (isotopologue 2); H2O 22 / 183 / 325 / 380 / ... GHz
    (+ H2-18O, H2-17O, HDO: isotopologues 2-4; speed-dependence parameter on 22 and 183 GHz); an O3 forest in clusters
    (isotopologues 1-5); the N2O and CO rotational ladders (isotopologues 1-3); a few 16O12C18O lines; HNO3 / SO2 / NO2 lines
    (molecules 9-12: in the file, beyond NMOL = 7); species-broadening data on `ibrd_frac` of the lines of molecules 1-7.
    Returns the records in file order and a dict of record indices of interest (first O2 coupling record of the 60-GHz
    complex etc.) for the block-layout of the fixture."""
    rng = np.random.default_rng(seed)
    rows = []

    def line(v, mol, iso, s, alfa, hwhm, epp, n, shift, sdep=0.0, iflg=0, lc=None):
        rows.append(dict(vnu=float(v), mol=mol, iso=iso, s=float(s), alfa=float(alfa), hwhm=float(hwhm), epp=float(epp), n=float(n),
                         shift=float(shift), sdep=float(sdep), iflg=iflg, lc=lc or []))

    def yg(y296, g296):
        y = y296 * np.array([1.42, 1.17, 1.0, 0.88])
        # a coupling record that the reference's LINES loop walks as if it were a line (after a mis-filed pair) takes its
        # isotopologue from the MOL word = the bits of Y(296 K) as REAL*4 (src/lnfl_mod.f90:67,80-82): nudge Y(296 K) by < 1e-4
        # relative so that those bits say isotopologue 1 - otherwise the reference reads scor(i,0), out of bounds
        b = int(np.float32(y[2]).view(np.int32))
        b += (150 - b % 1000) % 1000
        y[2] = float(np.int32(b).view(np.float32))
        return (y, g296 * np.array([1.9, 1.35, 1.0, 0.78]))

    # ---- O2
    line(1.0e-6 + 0.0002, 7, 1, 4e-33, 0.05, 0.05, 2.1, 0.8, 0.0, iflg=3, lc=[yg(0.05, 0.0)])      # non-resonant term
    for k, (fp, fm) in enumerate(zip(O2_NPLUS_GHZ, O2_NMINUS_GHZ)):
        N = 2 * k + 1
        s = 3.0e-25 * (2 * N + 1) * np.exp(-N * (N + 1) * 2.07 / 296.0) / 12.0   # rotational envelope, peak ~ N = 9
        w = 0.054 - 0.0007 * N
        for f, sign in ((fp, +1.0), (fm, -1.0)):
            y = sign * (0.25 - 0.035 * N + 0.0009 * N * N) * (1.0 + 0.03 * rng.standard_normal())
            g = -0.0007 * N * (1.0 + 0.1 * rng.standard_normal())
            line(f / GHZ_PER_WN, 7, 1, s * (1 + 0.05 * rng.standard_normal()), w, w, 1.4378 * N * (N + 1), 0.8, 0.0, iflg=1, lc=[yg(y, g)])
    for f in O2_SUBMM_GHZ:
        line(f / GHZ_PER_WN, 7, 1, 10 ** rng.uniform(-26.2, -25.2), 0.05, 0.05, rng.uniform(2, 600), 0.75, 0.0)
    for f in rng.uniform(50.0, 1600.0, 60):                                                             # 16O18O
        line(f / GHZ_PER_WN, 7, 2, 10 ** rng.uniform(-29.5, -28.0), 0.05, 0.05, rng.uniform(2, 900), 0.75, 0.0)
    # ---- H2O
    for i, f in enumerate(H2O_GHZ):
        strong = f in (556.9360, 752.0332, 987.9268, 1097.3648, 1113.3430, 1162.9116, 1207.6387, 1228.7888, 1410.6180)
        s = 10 ** (rng.uniform(-22.2, -21.0) if strong else rng.uniform(-24.9, -22.8))
        if f < 25:
            s = 4.4e-25
        if 183 < f < 184:
            s = 7.7e-23
        line(f / GHZ_PER_WN, 1, 1, s, rng.uniform(0.075, 0.105), rng.uniform(0.35, 0.52), rng.uniform(23, 1500), rng.uniform(0.55, 0.78),
             rng.uniform(-0.003, 0.001), sdep=(0.11 if f < 184 else 0.0))
    for iso, nl, ab in ((2, 18, 2e-3), (3, 10, 3.7e-4), (4, 30, 3.1e-4)):
        for f in rng.uniform(180.0, 1640.0, nl):
            line(f / GHZ_PER_WN, 1, iso, ab * 10 ** rng.uniform(-23.5, -21.3), rng.uniform(0.075, 0.105), rng.uniform(0.35, 0.52),
                 rng.uniform(20, 1200), rng.uniform(0.55, 0.78), rng.uniform(-0.003, 0.001))
    # ---- O3 forest: Q-branch-like clusters + a background
    centres = np.sort(rng.uniform(0.4, 54.6, 46))
    kinds = rng.choice(5, size=n_o3, p=[0.8, 0.08, 0.06, 0.03, 0.03]) + 1
    for j in range(n_o3):
        if rng.random() < 0.7:
            c = centres[rng.integers(0, len(centres))]
            v = abs(c + 0.16 * rng.standard_normal()) + 0.05
        else:
            v = rng.uniform(0.3, 54.9)
        iso = int(kinds[j])
        ab = (1.0, 4e-3, 2e-3, 7e-4, 4e-4)[iso - 1]
        line(min(v, 54.95), 3, iso, ab * 10 ** rng.uniform(-24.5, -20.8), rng.uniform(0.062, 0.084), rng.uniform(0.085, 0.11),
             rng.uniform(0, 1800), rng.uniform(0.62, 0.78), rng.uniform(-0.0015, 0.0005))
    # ---- N2O and CO ladders, 16O12C18O
    for iso, B, ab in ((1, 0.419011, 1.0), (2, 0.418982, 3.6e-3), (3, 0.404857, 3.6e-3)):
        for J in range(0, int(54.9 / (2 * B))):
            line(2 * B * (J + 1), 4, iso, ab * 1.2e-23 * (J + 1) ** 2 * np.exp(-B * J * (J + 1) * 1.4388 / 296.0) / 30.0, rng.uniform(0.07, 0.09),
                 rng.uniform(0.09, 0.11), B * J * (J + 1), 0.75, -0.0005)
    for iso, B, ab in ((1, 1.922529, 1.0), (2, 1.837972, 1.1e-2), (3, 1.830982, 2e-3)):
        for J in range(0, int(54.9 / (2 * B))):
            line(2 * B * (J + 1), 5, iso, ab * 3.3e-24 * (J + 1) ** 2 * np.exp(-B * J * (J + 1) * 1.4388 / 296.0), rng.uniform(0.055, 0.075),
                 rng.uniform(0.06, 0.08), B * J * (J + 1), 0.7, -0.0008)
    for J in range(2, 60, 3):
        line(0.7362 * (J + 1), 2, 3, 10 ** rng.uniform(-29.5, -28.0), 0.07, 0.09, 0.3681 * J * (J + 1), 0.7, -0.001)
    # ---- molecules beyond NMOL = 7 (real files hold them; LINES never visits them with NMOL = 7)
    for j in range(n_trace):
        mol = int(rng.choice([9, 10, 11, 12, 12, 12]))
        line(rng.uniform(0.3, 54.9), mol, 1, 10 ** rng.uniform(-24.0, -21.0), rng.uniform(0.07, 0.11), rng.uniform(0.07, 0.11) , rng.uniform(0, 900),
             rng.uniform(0.5, 0.75), 0.0)
    rows.sort(key=lambda r: r["vnu"])
    cols = {k: [] for k in ("vnu", "sp", "alfa", "epp", "mol", "hwhm", "tmpalf", "pshift", "iflg", "sdep")}
    brd_flg, brd_dat = [], []
    marks = {"o2_lc": [], "negepp": 0}
    for r in rows:
        v = r["vnu"]
        cols["vnu"].append(v); cols["sp"].append(r["s"] / (v * (1.0 - np.exp(-RADCN2 * v / 296.0)))); cols["alfa"].append(r["alfa"])
        cols["epp"].append(r["epp"]); cols["mol"].append(r["mol"] + 100 * r["iso"]); cols["hwhm"].append(r["hwhm"])
        cols["tmpalf"].append(r["n"]); cols["pshift"].append(r["shift"]); cols["iflg"].append(r["iflg"]); cols["sdep"].append(r["sdep"])
        flg, dat = [0] * 7, [0.0] * 21
        if r["mol"] <= 7 and rng.random() < ibrd_frac:
            flg = [int(x) for x in (rng.random(7) < 0.35)]
            for j in range(7):
                dat[3 * j: 3 * j + 3] = [rng.uniform(0.03, 0.15), rng.uniform(0.4, 0.8), rng.uniform(-0.004, 0.004)]
        brd_flg.append(flg); brd_dat.append(dat)
        for (y, g) in r["lc"]:
            if r["mol"] == 7 and r["iflg"] == 1:
                marks["o2_lc"].append(len(cols["vnu"]))
            cols["vnu"].append(y[0]); cols["sp"].append(g[0]); cols["alfa"].append(y[1]); cols["epp"].append(g[1])
            cols["mol"].append(int(np.float32(y[2]).view(np.int32))); cols["hwhm"].append(g[2]); cols["tmpalf"].append(y[3])
            cols["pshift"].append(g[3]); cols["iflg"].append(-r["iflg"]); cols["sdep"].append(0.0)
            brd_flg.append([0] * 7); brd_dat.append([0.0] * 21)
    rec = LineRecords(**{k: np.asarray(v) for k, v in cols.items()}, brd_flg=np.asarray(brd_flg), brd_dat=np.asarray(brd_dat))
    return rec, marks


def tune_epp_for_slot1_owner(epp: float, dbl_owner: int = 4, sgl_owner: int = 24) -> float:
    """The nearest REAL*4 value at or above `epp` for which the reference's owner rule for a coupling record that is the FIRST
    record of a block gives molecule `dbl_owner` in the "dbl" build and `sgl_owner` in the "sgl" build.
    Background: the reference takes that owner from bufr%mol(0), which aliases bufr%epp(250) (src/lnfl_mod.f90:50,
    struct_types.f90:45-58): MOD(K, 100) with K the bits of the REAL*4 value in the "sgl" build, and MOD of the bits of the
    REAL*8 widening = MOD(2^29 (K + 896 x 2^23), 100) in the "dbl" build - always a multiple of 4, so never O2, H2O, CO2 or O3:
    a straddled pair of those is always mis-filed by the default build (a real LNFL file cannot hold one)."""
    x = np.float32(epp)
    for _ in range(1000000):
        if slot1_owner(float(x), 8) == dbl_owner and slot1_owner(float(x), 4) == sgl_owner:
            return float(x)
        x = np.nextafter(x, np.float32(np.inf))
    raise RuntimeError("no such REAL*4 value nearby")


def slot1_owner(epp250: float, real_kind: int) -> int:
    """MOD(bufr%mol(0), 100) for a block whose 250th record holds lower-state energy `epp250` (see tune_epp_for_slot1_owner)."""
    x = np.float32(epp250)
    if real_kind == 4:
        b = int(x.view(np.int32))
    else:
        b = int(np.float64(x).view(np.int64))
    return int(np.fmod(b, 100))   # Fortran MOD: sign of the dividend


def real_like_file():
    """The line file of the real_like fixtures (tests/golden/make_golden.py) and of bench.py's c2real workload:
    realistic_lines() laid out like an LNFL product - second header record ('^'), 23 blocks, a short second block, a short
    mid-file block - plus ONE thing a real file cannot hold but the reference's reader defines: the coupling record of the O2
    118.75 GHz line as the FIRST record of a block (its line is the last record of the block before).  The reference then takes
    the record's owner from the bits of the block's 250th lower-state energy (tune_epp_for_slot1_owner): here N2O in the "dbl"
    build and molecule 24 in the "sgl" build - the O2 line is left without its record and LINES mis-walks the O2 list from
    there on (deterministic).  Returns (records, keyword arguments of tape3.write_tape3, indices of interest)."""
    rec, marks = realistic_lines()
    i = next(k for k in marks["o2_lc"] if abs(rec.vnu[k - 1] - 118.7503 / GHZ_PER_WN) < 1e-9)
    assert i > 260 and rec.iflg[i] == -1 and rec.iflg[i - 1] == 1
    j = i + 249                                   # the 250th record of the block that starts with the coupling record
    assert rec.iflg[j] >= 0 and rec.iflg[j + 1] >= 0, "pick another layout: the block must end on a plain line"
    rec.epp[j] = tune_epp_for_slot1_owner(float(rec.epp[j]))
    assert slot1_owner(float(rec.epp[j]), 8) == 4 and slot1_owner(float(rec.epp[j]), 4) == 24
    kw = dict(split_blocks_at=[i, i + 250, 2000, 2093], second_header=True, negepp_counts=([3, 0, 17], [3, 0, 17]))
    return rec, kw, dict(lc_slot1=i, tuned=j)


def sounder_channels() -> np.ndarray:
    """ATMS / AMSU-like channel centres and passband offsets [cm-1], ascending (40 values)."""
    ghz = [22.235, 23.8, 31.4, 36.5, 50.3, 51.76, 52.8, 53.246, 53.596, 53.948, 54.4, 54.94, 55.5, 56.02, 56.5, 57.07, 57.2903,
           57.5073, 57.612, 57.9, 58.8, 60.79, 63.28, 88.2, 89.0, 118.0, 118.55, 118.7503, 118.95, 120.5, 150.0, 165.5, 176.31, 180.31,
           182.31, 183.3101, 184.31, 186.31, 190.31, 229.0]
    return np.sort(np.array(ghz) / GHZ_PER_WN)
