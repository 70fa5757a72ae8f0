"""TAPE3 binary line-file format (LNFL output as read by monoRTM).

Python writer/reader used to build synthetic line files for tests and benchmarks; the
product's own parser is C++ (monortm_amd/csrc/tape3.cpp).  Layout follows what the
reference reads: file header ``src/lnfl_mod.f90:250-252``, panel header + 250-slot
field-major line block ``src/struct_types.f90:27-43`` / ``src/lnfl_mod.f90:157-168``,
Fortran sequential-unformatted framing with 4-byte record markers
(``build/makefile.common:198`` ``-frecord-marker=4``).
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field

import numpy as np

NLINEREC = 250  # slots per block, src/struct_types.f90:27
MXBRDMOL = 7
BLOCK_WORDS = 9750  # 4-byte words in one line block


@dataclass
class LineRecords:
    """Raw TAPE3 records, one entry per 156-byte slot, in file order.

    A physical line has ``iflg >= 0``; ``iflg`` in {1,3,5} announces that the next
    record(s) carry line-coupling coefficients and have ``iflg`` in {-1,-3,-5}
    (``src/lnfl_mod.f90:46-64``).  In such a record the eight fields
    vnu,sp,alfa,epp,mol(bit-cast f32),hwhm,tmpalf,pshift hold Y/G at 200/250/296/340 K
    (``src/modm.f90:331-338``).
    """

    vnu: np.ndarray      # f64
    sp: np.ndarray       # f32  S / (nu (1-exp(-c2 nu/296)))
    alfa: np.ndarray     # f32  air/foreign HWHM
    epp: np.ndarray      # f32  lower-state energy
    mol: np.ndarray      # i32  molecule + 100*isotope   (or f32 bits in an LC record)
    hwhm: np.ndarray     # f32  self HWHM
    tmpalf: np.ndarray   # f32  temperature exponent
    pshift: np.ndarray   # f32  pressure shift
    iflg: np.ndarray     # i32
    brd_flg: np.ndarray = field(default=None)   # i32 [n,7]
    brd_dat: np.ndarray = field(default=None)   # f32 [n,21] (hw,tmp,shift)x7
    sdep: np.ndarray = field(default=None)      # f32

    def __post_init__(self):
        n = len(self.vnu)
        self.vnu = np.asarray(self.vnu, np.float64)
        for k in ("sp", "alfa", "epp", "hwhm", "tmpalf", "pshift"):
            setattr(self, k, np.asarray(getattr(self, k), np.float32))
        self.mol = np.asarray(self.mol, np.int32)
        self.iflg = np.asarray(self.iflg, np.int32)
        if self.brd_flg is None:
            self.brd_flg = np.zeros((n, MXBRDMOL), np.int32)
        if self.brd_dat is None:
            self.brd_dat = np.zeros((n, 3 * MXBRDMOL), np.float32)
        if self.sdep is None:
            self.sdep = np.zeros(n, np.float32)
        self.brd_flg = np.asarray(self.brd_flg, np.int32).reshape(n, MXBRDMOL)
        self.brd_dat = np.asarray(self.brd_dat, np.float32).reshape(n, 3 * MXBRDMOL)
        self.sdep = np.asarray(self.sdep, np.float32)

    def __len__(self):
        return len(self.vnu)

    @property
    def n_physical(self) -> int:
        return int(np.count_nonzero(self.iflg >= 0))


def _rec(payload: bytes) -> bytes:
    m = struct.pack("<i", len(payload))
    return m + payload + m


def _pad8(s: str) -> bytes:
    return s.encode("ascii")[:8].ljust(8)


def write_tape3(path: str, rec: LineRecords, split_blocks_at: list[int] | None = None, second_header: bool = False,
                keep_pairs: bool = True, negepp_counts: tuple | None = None) -> None:
    """Write ``rec`` as a TAPE3 file.  Records must already be in file order
    (ascending vnu for physical lines, LC records right after their line).
    ``split_blocks_at`` optionally forces block boundaries at the given record indices
    (to exercise the reader's block skip / stop logic; two boundaries close together make a
    short mid-file block).  ``second_header``: character 8 of HLINID(7) is '^' and a second
    header record (N_NEGEPP(64), N_RESETEPP(64), XSPACE(4096): ``src/lnfl_mod.f90:258-262``)
    follows the first, as LNFL writes it when it met negative lower-state energies - every
    aer_v_3.x line file carries one.  ``keep_pairs=False``: a forced boundary may separate a
    line from its coupling record (the record is then slot 1 of the next block)."""
    n = len(rec)
    hlinid = [_pad8("SYNTH"), _pad8("LNFL"), _pad8(""), _pad8(""), _pad8(""), _pad8(""),
              _pad8("       ^" if second_header else "       "), _pad8(""), _pad8(""), _pad8("LNFL 91I")]  # char 8 of #10 == 'I'
    bmolid = [_pad8("")] * 64
    hdr = b"".join(hlinid) + b"".join(bmolid)
    molcnt = np.zeros(64, np.int32)
    phys = rec.iflg >= 0
    for m in np.unique(rec.mol[phys] % 100):
        if 1 <= m <= 64:
            molcnt[m - 1] = np.count_nonzero(rec.mol[phys] % 100 == m)
    hdr += molcnt.tobytes() + np.zeros(64, np.int32).tobytes() + np.zeros(64, np.int32).tobytes()
    hdr += np.zeros(64, np.float32).tobytes()
    vlo = float(rec.vnu[phys].min()) if phys.any() else 0.0
    vhi = float(rec.vnu[phys].max()) if phys.any() else 0.0
    hdr += struct.pack("<iffiiiii", 39, vlo, vhi, int(phys.sum()), 0, 0, 0, 0)
    hdr += _pad8("") * 2
    assert len(hdr) == 1664

    bounds = sorted(set([0, n] + list(split_blocks_at or [])))
    starts = []
    for a, b in zip(bounds[:-1], bounds[1:]):
        s = a
        while s < b:
            e = min(s + NLINEREC, b)
            # never separate a line from its coupling records (the reader looks at
            # bufr%mol(ik-1) inside one block, src/lnfl_mod.f90:50-58)
            while keep_pairs and e < b and e > s + 1 and rec.iflg[e] < 0:
                e -= 1
            starts.append((s, e))
            s = e
    with open(path, "wb") as f:
        f.write(_rec(hdr))
        if second_header:
            nneg = np.zeros(64, np.int32)
            nres = np.zeros(64, np.int32)
            if negepp_counts:
                nneg[: len(negepp_counts[0])] = negepp_counts[0]
                nres[: len(negepp_counts[1])] = negepp_counts[1]
            f.write(_rec(nneg.tobytes() + nres.tobytes() + np.zeros(4096, np.float32).tobytes()))
        for s, e in starts:
            k = e - s
            pv = rec.vnu[s:e][rec.iflg[s:e] >= 0]
            vmin = float(pv.min()) if len(pv) else 0.0
            vmax = float(pv.max()) if len(pv) else 0.0
            f.write(_rec(struct.pack("<ddii", vmin, vmax, k, BLOCK_WORDS)))

            def padded(a, dt, width=None):
                shape = (NLINEREC,) if width is None else (NLINEREC, width)
                out = np.zeros(shape, dt)
                out[:k] = a[s:e]
                return out.tobytes()

            blk = (padded(rec.vnu, np.float64) + padded(rec.sp, np.float32)
                   + padded(rec.alfa, np.float32) + padded(rec.epp, np.float32)
                   + padded(rec.mol, np.int32) + padded(rec.hwhm, np.float32)
                   + padded(rec.tmpalf, np.float32) + padded(rec.pshift, np.float32)
                   + padded(rec.iflg, np.int32) + padded(rec.brd_flg, np.int32, MXBRDMOL)
                   + padded(rec.brd_dat, np.float32, 3 * MXBRDMOL) + padded(rec.sdep, np.float32))
            assert len(blk) == 4 * BLOCK_WORDS
            f.write(_rec(blk))


def read_tape3(path: str) -> LineRecords:
    """Read every record of every block (no v1/v2 filtering)."""
    data = open(path, "rb").read()
    pos = 0

    def rec():
        nonlocal pos
        (m,) = struct.unpack_from("<i", data, pos)
        payload = data[pos + 4: pos + 4 + m]
        pos += m + 8
        return payload

    hdr = rec()
    if chr(hdr[7 * 8 - 1 + 0]) == "^":  # char 8 of HLINID(7): second header record follows
        rec()
    cols = {k: [] for k in ("vnu", "sp", "alfa", "epp", "mol", "hwhm", "tmpalf", "pshift", "iflg",
                            "brd_flg", "brd_dat", "sdep")}
    while pos < len(data):
        ph = rec()
        vmin, vmax, nrec, nwds = struct.unpack("<ddii", ph[:24])
        blk = rec()
        off = 0

        def take(dt, count):
            nonlocal off
            a = np.frombuffer(blk, dt, count, off)
            off += a.nbytes
            return a

        cols["vnu"].append(take(np.float64, NLINEREC)[:nrec])
        for k in ("sp", "alfa", "epp"):
            cols[k].append(take(np.float32, NLINEREC)[:nrec])
        cols["mol"].append(take(np.int32, NLINEREC)[:nrec])
        for k in ("hwhm", "tmpalf", "pshift"):
            cols[k].append(take(np.float32, NLINEREC)[:nrec])
        cols["iflg"].append(take(np.int32, NLINEREC)[:nrec])
        cols["brd_flg"].append(take(np.int32, NLINEREC * MXBRDMOL).reshape(NLINEREC, MXBRDMOL)[:nrec])
        cols["brd_dat"].append(take(np.float32, NLINEREC * 21).reshape(NLINEREC, 21)[:nrec])
        cols["sdep"].append(take(np.float32, NLINEREC)[:nrec])
    return LineRecords(**{k: np.concatenate(v) for k, v in cols.items()})
