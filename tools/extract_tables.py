#!/usr/bin/env python3
"""Extract the numerical model tables (MT_CKD continuum coefficients, TIPS-2003 partition
sums, isotopologue masses) from the reference's DATA statements into a C header.

The tables are the *definition* of the physical model (AER MT_CKD_3.5, Gamache TIPS 2003);
only numbers are extracted - no executable statements.  Run in the build container only
(needs /root/reference); the generated header is committed because the GPU box has no
reference tree.

    python tools/extract_tables.py [/root/reference] > monortm_amd/csrc/tables/monortm_tables.h

Sources: src/contnm.f90 BLOCK DATA units (:1473 BS296, :1981 BS260, :2489 BFH2O, :3018 BFCO2,
:4232 BN2T296, :4281 BN2T220, ...), src/tips_2003.f90 QT_* routines + BDtdat (:312) +
ISO_2002 (:372), src/isotope.incl (:10-167), src/contnm.f90:174-202 (XFAC_RHU, XFACREV),
:89-172 (XFACCO2), src/contnm.f90:2963-2975 (tdep_bandhead).
"""
from __future__ import annotations

import re
import sys
from collections import OrderedDict


def logical_lines(path: str, fixed_incl: bool = False):
    """Yield free-form logical statements with continuations joined and comments removed."""
    out = []
    cur = ""
    cont = False
    for raw in open(path, encoding="latin-1"):
        line = raw.rstrip("\n")
        # strip comment (no '!' inside the string literals we care about)
        if "!" in line:
            line = line[: line.index("!")]
        line = line.rstrip()
        if not line.strip():
            continue
        s = line.strip()
        if cont or s.startswith("&"):
            if s.startswith("&"):
                s = s[1:]
        else:
            if cur:
                out.append(cur)
            cur = ""
        cont = s.endswith("&")
        if cont:
            s = s[:-1]
        cur += " " + s
    if cur:
        out.append(cur)
    return [c.strip() for c in out]


NUM = r"[+-]?(?:\d+\.?\d*|\.\d+)(?:[EeDd][+-]?\d+)?"


def parse_values(txt: str):
    vals = []
    for tok in txt.split(","):
        tok = tok.strip()
        if not tok:
            continue
        m = re.fullmatch(r"(\d+)\s*\*\s*(" + NUM + ")", tok)
        if m:
            vals += [m.group(2)] * int(m.group(1))
        else:
            if not re.fullmatch(NUM, tok):
                raise ValueError(f"bad numeric token {tok!r}")
            vals.append(tok)
    return [float(v.replace("D", "E").replace("d", "e")) for v in vals]


def split_data_stmt(stmt: str):
    """'DATA a,b / 1,2 /, c / 3 /' -> [(['a','b'], [1,2]), (['c'],[3])]"""
    body = stmt[4:].strip()
    res = []
    while body:
        i = body.index("/")
        j = body.index("/", i + 1)
        names = body[:i].strip().strip(",").strip()
        vals = parse_values(body[i + 1: j])
        res.append((names, vals))
        body = body[j + 1:].strip()
        if body.startswith(","):
            body = body[1:].strip()
    return res


def split_top(s: str):
    parts, depth, cur = [], 0, ""
    for ch in s:
        if ch == "(":
            depth += 1
        elif ch == ")":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        parts.append(cur.strip())
    return parts


def parse_block_data(stmts):
    """Return {common_name: OrderedDict(var -> list of floats)} for one BLOCK DATA unit."""
    commons = OrderedDict()
    sizes = {}
    for st in stmts:
        m = re.match(r"(?i)common\s*/\s*(\w+)\s*/(.*)", st)
        if m:
            od = commons.setdefault(m.group(1).upper(), OrderedDict())
            for v in split_top(m.group(2)):
                mm = re.match(r"(\w+)\s*(?:\(\s*(\d+)\s*\))?$", v)
                if not mm:
                    raise ValueError(v)
                name = mm.group(1).upper()
                sizes[name] = int(mm.group(2)) if mm.group(2) else 1
                od[name] = None
    data = {}
    for st in stmts:
        if re.match(r"(?i)data\b", st):
            for names, vals in split_data_stmt(st):
                nl = split_top(names)
                if len(nl) == 1 and not nl[0].startswith("("):
                    data[nl[0].upper()] = vals
                elif all(re.fullmatch(r"\w+", n) for n in nl):
                    assert len(nl) == len(vals), (nl, len(vals))
                    for n, v in zip(nl, vals):
                        data[n.upper()] = [v]
                else:
                    # implied DO over a sub-range: (NAME(I), I = a, b) fills elements a..b
                    mm = re.fullmatch(r"\(\s*(\w+)\s*\(\s*\w+\s*\)\s*,\s*\w+\s*=\s*(\d+)\s*,\s*(\d+)\s*\)", names)
                    if not mm:
                        raise ValueError(names)
                    nm, a, b = mm.group(1).upper(), int(mm.group(2)), int(mm.group(3))
                    assert len(vals) == b - a + 1, (nm, a, b, len(vals))
                    arr = data.setdefault(nm, [None] * sizes.get(nm, b))
                    if len(arr) < b:
                        arr.extend([None] * (b - len(arr)))
                    arr[a - 1:b] = vals
    for k, v in data.items():
        if any(x is None for x in v):
            raise ValueError(f"{k}: DATA statements leave holes")
    for cname, od in commons.items():
        for v in od:
            if v in data:
                if len(data[v]) != sizes[v]:
                    raise ValueError(f"{cname}:{v} has {len(data[v])} values, declared {sizes[v]}")
                od[v] = data[v]
    return commons


def units(stmts, start_re, end_re):
    cur, name = None, None
    for st in stmts:
        m = re.match(start_re, st)
        if m and cur is None:
            cur, name = [], m.group(1)
            continue
        if cur is not None:
            if re.match(end_re, st):
                yield name, cur
                cur = None
            else:
                cur.append(st)


def fmt(vals, per=6):
    lines = []
    for i in range(0, len(vals), per):
        lines.append("  " + ", ".join(repr(float(v)) for v in vals[i:i + per]) + ",")
    return "\n".join(lines)


def emit_array(name, vals, ctype="double"):
    if ctype == "int":
        body = "  " + ", ".join(str(int(v)) for v in vals)
    else:
        body = fmt(vals)
    return f"static const {ctype} {name}[{len(vals)}] = {{\n{body}\n}};\n"


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    out = []
    w = out.append
    w("/* GENERATED by tools/extract_tables.py from the reference's DATA statements - do not edit.\n"
      " *\n"
      " * Numerical model tables only (no code): MT_CKD_3.5 continuum coefficients\n"
      " * (src/contnm.f90; Copyright Atmospheric & Environmental Research, Inc. (AER); the notice at\n"
      " * src/contnm.f90:6-22 permits use and redistribution for scientific and research purposes\n"
      " * provided the notice is kept and acknowledgment is given to AER), TIPS-2003 total internal\n"
      " * partition sums (Fischer, Gamache, Goldman, Rothman, Perrin, JQSRT 2003; src/tips_2003.f90)\n"
      " * and HITRAN isotopologue masses (src/isotope.incl).\n"
      " * Every value is the literal decimal string of the reference parsed as an IEEE double, which\n"
      " * is what the reference's \"dbl\" build (-fdefault-real-8) holds.\n"
      " */\n#ifndef MONORTM_TABLES_H\n#define MONORTM_TABLES_H\n")

    # ---------------- continuum ----------------
    st = logical_lines(f"{ref}/src/contnm.f90")
    want = {
        "SH2O": "MT_SELF296", "S260": "MT_SELF260", "FH2O": "MT_FRGN296", "FCO2": "MT_FCO2",
        "N2RT296": "MT_N2RT296", "N2RT220": "MT_N2RT220",
    }
    found = {}
    for bname, body in units(st, r"(?i)block\s*data\s+(\w+)", r"(?i)end(\s+block\s*data.*)?$"):
        try:
            cm = parse_block_data(body)
        except Exception as e:  # tables we do not extract may use constructs we do not parse
            sys.stderr.write(f"skip BLOCK DATA {bname}: {e}\n")
            continue
        for cname, od in cm.items():
            found[cname] = od
    for cname, cid in want.items():
        od = found[cname]
        keys = list(od.keys())
        v1, v2, dv, npt = (od[k][0] for k in keys[:4])
        w(f"/* COMMON /{cname}/ : grid v1={v1} v2={v2} dv={dv} npt={int(npt)} */\n")
        w(f"#define {cid}_V1 {v1!r}\n#define {cid}_V2 {v2!r}\n#define {cid}_DV {dv!r}\n"
          f"#define {cid}_NPT {int(npt)}\n")
        if cname.startswith("N2RT"):
            w(emit_array(cid + "_C", od[keys[4]]))
            w(emit_array(cid + "_SF", od[keys[5]]))
        else:
            flat = []
            for k in keys[4:]:
                flat += od[k]
            assert len(flat) == int(npt), (cname, len(flat), npt)
            w(emit_array(cid, flat))

    # ---- branches that only act above 1340 cm-1 (O2 / N2 fundamentals and overtone, O3 Chappuis / Hartley-Huggins,
    # O2 near-IR, visible and UV)
    def emit_grid(cname, cid):
        od = found[cname]
        keys = list(od.keys())
        v1, v2, dv, npt = (od[k][0] for k in keys[:4])
        w(f"/* COMMON /{cname}/ : grid v1={v1} v2={v2} dv={dv} npt={int(npt)} */\n")
        w(f"#define {cid}_V1 {v1!r}\n#define {cid}_V2 {v2!r}\n#define {cid}_DV {dv!r}\n#define {cid}_NPT {int(npt)}\n")
        return od, keys[4:], int(npt)

    for cname, cid, parts in (("O3CHAP", "MT_O3CH", ("X", "Y", "Z")), ("N2_F", "MT_N2F", ("272", "228", "AH2O")),
                              ("O2_F", "MT_O2F", ("XO2", "XO2T"))):
        od, keys, npt = emit_grid(cname, cid)
        if len(keys) == len(parts):                       # separate named arrays
            for k, suffix in zip(keys, parts):
                assert len(od[k]) == npt, (cname, k, len(od[k]))
                w(emit_array(f"{cid}_{suffix}", od[k]))
        else:                                             # one flat run of sub-blocks per part
            flat = []
            for k in keys:
                flat += od[k]
            assert len(flat) == npt * len(parts), (cname, len(flat), npt)
            for i, suffix in enumerate(parts):
                w(emit_array(f"{cid}_{suffix}", flat[i * npt:(i + 1) * npt]))
    for cname, cid in (("O3HH0", "MT_O3HH0"), ("O3HH1", "MT_O3HH1"), ("O3HH2", "MT_O3HH2"), ("O3HUV", "MT_O3HUV"),
                       ("N2_F1", "MT_N2F1"), ("O2INF1_MATE", "MT_O2INF1"), ("O2INF3_ABAND", "MT_O2INF3"),
                       ("O2_O2_VIS", "MT_O2VIS"), ("O2_FUV", "MT_O2FUV")):
        od, keys, npt = emit_grid(cname, cid)
        flat = []
        for k in keys:
            flat += od[k]
        assert len(flat) == npt, (cname, len(flat), npt)
        w(emit_array(cid, flat))

    # in-routine DATA of CONTNM / FRNCO2
    joined = {}
    for s in st:
        if re.match(r"(?i)data\b", s):
            try:
                parsed = split_data_stmt(s)
            except ValueError:
                continue  # character DATA etc.
            for names, vals in parsed:
                joined.setdefault(names.replace(" ", "").upper(), vals)
    w("/* src/contnm.f90:186-202  XFAC_RHU(-1:61) foreign-continuum scaling, 10 cm-1 bins */\n")
    w(emit_array("MT_XFAC_RHU", joined["(XFAC_RHU(I),I=-1,61)"]))
    w("/* src/contnm.f90:177-180  XFACREV(0:14) */\n")
    w(emit_array("MT_XFACREV", joined["(XFACREV(I),I=0,14)"]))
    w("/* src/contnm.f90:92-172  XFACCO2(500) */\n")
    w(emit_array("MT_XFACCO2", joined["XFACCO2"]))
    w("/* src/contnm.f90:2969-2975  tdep_bandhead(1196:1220), t_eff */\n")
    w(emit_array("MT_TDEP_BANDHEAD", joined["(TDEP_BANDHEAD(I),I=1196,1220)"]))

    # ---------------- TIPS ----------------
    st = logical_lines(f"{ref}/src/tips_2003.f90")
    tdat = None
    isonm = None
    for s in st:
        if re.match(r"(?i)data\s+tdat\s*/", s):
            tdat = split_data_stmt(s)[0][1]
        if re.match(r"(?i)data\s*\(\s*isonm", s):
            isonm = [int(v) for v in split_data_stmt(s)[0][1]]
    assert len(tdat) == 119 and len(isonm) == 39
    w("/* src/tips_2003.f90:312-336  temperature grid */\n")
    w(emit_array("TIPS_TDAT", tdat))
    w("/* src/tips_2003.f90:380-389  ISONM: isotopologues per molecule known to TIPS */\n")
    w(emit_array("TIPS_ISONM", isonm, "int"))
    # QT routines in call order of TIPS_2003 (src/tips_2003.f90:64-270): molecule index -> routine
    order = ["H2O", "CO2", "O3", "N2O", "CO", "CH4", "O2", "NO", "SO2", "NO2", "NH3", "HNO3", "OH",
             "HF", "HCL", "HBR", "HI", "CLO", "OCS", "H2CO", "HOCL", "N2", "HCN", "CH3CL", "H2O2",
             "C2H2", "C2H6", "PH3", "COF2", "SF6", "H2S", "HCOOH", "HO2", "O", "CLONO2", "NOP",
             "HOBR", "C2H4"]
    qts = {}
    for rname, body in units(st, r"(?i)subroutine\s+QT_(\w+)", r"(?i)end$"):
        rows = {}
        for s in body:
            if re.match(r"(?i)data\s*\(\s*QofT", s):
                for names, vals in split_data_stmt(s):
                    mm = re.match(r"(?i)\(\s*QofT\s*\(\s*(\d+)\s*,\s*J\s*\)", names)
                    rows[int(mm.group(1))] = vals
        qts[rname.upper()] = rows
    offsets = []
    flat = []
    for mi, nm in enumerate(order):
        rows = qts[nm]
        n = len(rows)
        assert sorted(rows) == list(range(1, n + 1)), nm
        assert n == isonm[mi] or nm == "O", (nm, n, isonm[mi])
        offsets.append(len(flat) // 119)
        for i in range(1, n + 1):
            assert len(rows[i]) == 119
            flat += rows[i]
    offsets.append(len(flat) // 119)
    w("/* TIPS_QOFT[(TIPS_OFFSET[mol-1] + iso-1)*119 + j] = QofT(iso, j) of routine QT_<mol>;\n"
      "   molecules 1..38 (39 = CH3OH uses the classical formula, src/tips_2003.f90:262-270;\n"
      "   34 = O has Q=1, src/tips_2003.f90:235-240) */\n")
    w(emit_array("TIPS_OFFSET", offsets, "int"))
    w(emit_array("TIPS_QOFT", flat))

    # ---------------- isotope masses ----------------
    st = logical_lines(f"{ref}/src/isotope.incl")
    smass = [[0.0] * 9 for _ in range(39)]
    iso_max = None
    for s in st:
        if re.match(r"(?i)data\b", s):
            for names, vals in split_data_stmt(s):
                mm = re.match(r"(?i)\(\s*smass\s*\(\s*(\d+)\s*,\s*i\s*\)\s*,\s*i\s*=\s*1\s*,\s*(\d+)\s*\)", names)
                if mm:
                    m_, n_ = int(mm.group(1)), int(mm.group(2))
                    assert len(vals) == n_
                    smass[m_ - 1][:n_] = vals
                if re.match(r"(?i)\(\s*iso_max", names):
                    iso_max = [int(v) for v in vals]
    assert iso_max and len(iso_max) == 39
    w("/* src/isotope.incl:16-24  ISO_MAX(39) */\n")
    w(emit_array("ISO_MAX", iso_max, "int"))
    w("/* src/isotope.incl:51-167  SMASS(mol,iso) g/mol, row-major [39][9] */\n")
    w(emit_array("ISO_SMASS", [v for row in smass for v in row]))
    w("#endif\n")
    sys.stdout.write("".join(out))


if __name__ == "__main__":
    main()
