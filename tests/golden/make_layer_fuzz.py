#!/usr/bin/env python3
"""Layering fixtures for the IATM = 1 front end (monortm_amd/fortran/lblatm_front.f90): seeded random decks over the six model
atmospheres, the path cases 2A / 3A / 3B, automatic and given layering (altitudes or pressures), molecule counts and the
NOZERO flag, each run through the reference PROGRAM MONORTM (oracle/_ref/monortm_ref_dbl, IPUNCH = 1) for its TAPE7.
Stores MONORTM.IN + TAPE7.expected under tests/golden/layers_fuzz/NN/.  Decks the reference itself refuses are dropped.
Only runs where the compiled reference exists."""
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
EXE = os.path.join(ROOT, "oracle", "_ref", "monortm_ref_dbl")
T3 = os.path.join(ROOT, "tests", "golden", "decks", "TAPE3_synthetic")
OUT = os.path.join(ROOT, "tests", "golden", "layers_fuzz")
HEAD = open(os.path.join(ROOT, "tests", "golden", "decks", "case1_MDL_ATM_dn", "MONORTM.IN")).read().split("\n")
K31 = next(i for i, ln in enumerate(HEAD) if ln.split()[:7] == ["6", "2", "0", "1", "1", "22", "1"])
ZTOP = {1: 100.0, 2: 100.0, 3: 100.0, 4: 100.0, 5: 100.0, 6: 100.0}


def deck(rng):
    model = int(rng.integers(1, 7))
    nmol = int(rng.choice([7, 12, 22, 28]))
    nozero = int(rng.integers(0, 2))
    kind = rng.choice(["up", "down", "space", "limb", "pbnd"], p=[0.3, 0.25, 0.2, 0.15, 0.1])
    itype, h1, h2, ang = 2, 0.0, 0.0, 0.0
    if kind == "up":
        h1, h2, ang = rng.uniform(0, 3), rng.uniform(20, 95), rng.uniform(0, 80)
    elif kind == "down":
        h1, h2, ang = rng.uniform(20, 95), rng.uniform(0, 3), rng.uniform(110, 180)
    elif kind == "space":
        itype, h1, ang = 3, rng.uniform(0, 5), rng.uniform(0, 85)
    elif kind == "limb":
        itype, h1, h2 = 3, rng.uniform(30, 95), rng.uniform(5, 25)
    lines = list(HEAD[:K31])
    rec33 = []
    if kind == "pbnd":
        pb = np.sort(rng.uniform(15.0, 900.0, int(rng.integers(6, 20))))[::-1]
        psfc = {1: 1013.0, 2: 1013.0, 3: 1018.0, 4: 1010.0, 5: 1013.0, 6: 1013.0}[model]
        pb = np.concatenate([[psfc], pb])
        ibmax = -len(pb)
        h1, h2, ang = pb[0], pb[-1], rng.uniform(0, 60)
        rec33 = ["".join(f"{v:10.3f}" for v in pb[i:i + 8]) for i in range(0, len(pb), 8)]
    elif rng.random() < 0.5:
        ibmax = 0
        rec33 = ["".join(f"{v:10.3f}" for v in (rng.choice([0.0, 1.3, 2.0]), rng.choice([0.0, 3.0, 7.0]), rng.choice([0.0, 10.0, 15.0]),
                                               0.0, 0.0))]
    else:
        lo, hi = (min(h1, h2), max(h1, h2)) if itype == 2 else (h2 if kind == "limb" else h1, 100.0)
        zb = np.unique(np.round(np.concatenate([[lo, hi], rng.uniform(lo, hi, int(rng.integers(8, 30)))]), 3))
        ibmax = len(zb)
        rec33 = ["".join(f"{v:10.3f}" for v in zb[i:i + 8]) for i in range(0, len(zb), 8)]
    lines.append(f"{model:5d}{itype:5d}{ibmax:5d}{nozero:5d}{1:5d}{nmol:5d}{1:5d}")
    lines.append(f"{h1:10.3f}{h2:10.3f}{ang:10.3f}")
    lines += rec33
    lines += HEAD[K31 + 3:]
    return "\n".join(lines), kind


if __name__ == "__main__":
    rng = np.random.default_rng(20261004)
    os.makedirs(OUT, exist_ok=True)
    kept, tried = 0, 0
    want = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    while kept < want and tried < 4 * want:
        tried += 1
        text, kind = deck(rng)
        with tempfile.TemporaryDirectory() as w:
            open(os.path.join(w, "MONORTM.IN"), "w").write(text)
            shutil.copy(T3, os.path.join(w, "TAPE3"))
            try:
                r = subprocess.run([EXE], cwd=w, capture_output=True, text=True, timeout=300)
            except subprocess.TimeoutExpired:
                continue
            if r.returncode != 0 or not os.path.exists(os.path.join(w, "TAPE7")) or os.path.getsize(os.path.join(w, "TAPE7")) < 200:
                print("dropped", kind, (r.stdout + r.stderr)[-120:].replace("\n", " "))
                continue
            d = os.path.join(OUT, f"{kept:02d}_{kind}")
            os.makedirs(d, exist_ok=True)
            open(os.path.join(d, "MONORTM.IN"), "w").write(text)
            shutil.copy(os.path.join(w, "TAPE7"), os.path.join(d, "TAPE7.expected"))
            kept += 1
            print("kept", d)
