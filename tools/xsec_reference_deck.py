#!/usr/bin/env python3
"""What does the reference PROGRAM do with IXSECT = 1?  Build an IATM = 0 deck with cross-section molecules from example case 4
(MONORTM.IN with the XSCT flag set and infrared channels, MONORTM_PROF.IN extended by records 2.2.x as src/monortm.f90:492-530
reads them, a synthetic FSCDXS / xs library) in a scratch directory and run oracle/_ref/monortm_ref_dbl on it.

    python tools/xsec_reference_deck.py [scratch dir]        (build container only: needs oracle/_ref)

Finding (round 3, flang -O0 build): see DESIGN.md section 7 - the driver allocates ODXSEC(nwn, mxlay) while MONORTM_XSEC_SUB
indexes it as (NWNMX, MXLAY) (src/monortm_sub.F90:1611), so every layer but the first is written outside the array.
"""
import os
import shutil
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from monortm_amd import xsec  # noqa: E402


def build_deck(d: str, ixsect: int = 1):
    """Write MONORTM.IN, MONORTM_PROF.IN (with records 2.2.x), TAPE3, FSCDXS and the xs files into d.
    -> (wn [4], P [nlay], T [nlay], xamnt [nlay, 3], names)."""
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d)
    src = os.path.join(ROOT, "tests", "golden", "decks", "case4_IATM0_dn")
    mi = open(os.path.join(src, "MONORTM.IN")).read().split("\n")
    k = next(i for i, ln in enumerate(mi) if ln.startswith("$ Rundeck")) + 1
    ln = mi[k].ljust(90)
    mi[k] = (ln[:69] + str(ixsect) + ln[70:]).rstrip()        # XSCT (column 70, src/monortm_sub.F90:884)
    wn = [785.0, 795.0, 846.0, 921.0]
    n_old = int(mi[k + 2])
    mi[k + 2:k + 3 + n_old] = [str(len(wn))] + [f"{w:.6f}" for w in wn]
    open(os.path.join(d, "MONORTM.IN"), "w").write("\n".join(mi))
    prof = open(os.path.join(src, "MONORTM_PROF.IN")).read().rstrip("\n").split("\n")
    nlay = int(prof[0][2:5])
    names = xsec.synthetic_library(d, f12_pres_mb=20.0)
    # records 2.2: IXMOLS, IXSBIN (I5,5X,I5); names (7A10); 2.2.3 header (1X,I1,I3,I5,F10.2,15A4); per layer the layer record
    # (format 910 / 915) and the amounts (8E15.7): XAMNT(1:7), WBRODX
    out = list(prof)
    out.append(f"{len(names):5d}     {0:5d}")
    out.append("".join(f"{n:<10s}" for n in names))
    out.append(f" 1{nlay:3d}{len(names):5d}{1.0:10.2f} synthetic cross-section amounts")
    P, T, XA = [], [], []
    for il in range(nlay):
        rec = prof[1 + il * 4]
        p, t = float(rec[:15]), float(rec[15:25])
        P.append(p)
        T.append(t)
        if il == 0:
            out.append(f"{p:15.7E}{t:10.4f}{1.0:10.4f}   {0:2d} " + f"{0.0:7.2f}{1013.0:8.3f}{288.2:7.2f}" + f"{0.7:7.2f}{931.6:8.3f}{283.6:7.2f}")
        else:
            out.append(f"{p:15.7E}{t:10.4f}{1.0:10.4f}   {0:2d}" + " " * 23 + f"{0.7 * (il + 1):7.2f}{p:8.3f}{t:7.2f}")
        amt = [3.0e14 * np.exp(-il / 6.0), 6.0e14 * np.exp(-il / 6.0), 1.2e15 * np.exp(-il / 6.0)]
        XA.append([float(f"{x:15.7E}") for x in amt])
        out.append("".join(f"{x:15.7E}" for x in amt + [0.0] * 4 + [1.0e24]))
    open(os.path.join(d, "MONORTM_PROF.IN"), "w").write("\n".join(out if ixsect else prof) + "\n")
    shutil.copy(os.path.join(ROOT, "tests", "golden", "decks", "TAPE3_synthetic"), os.path.join(d, "TAPE3"))
    os.makedirs(os.path.join(d, "in"), exist_ok=True)
    return np.array(wn), np.array(P), np.array(T), np.array(XA), names


def main():
    d = sys.argv[1] if len(sys.argv) > 1 else "/tmp/monortm_xs_deck"
    build_deck(d)

    def big_stack():
        import resource
        resource.setrlimit(resource.RLIMIT_STACK, (resource.RLIM_INFINITY, resource.RLIM_INFINITY))
    exe = os.path.join(ROOT, "oracle", "_ref", "monortm_ref_dbl")
    r = subprocess.run([exe], cwd=d, capture_output=True, text=True, preexec_fn=big_stack, timeout=600)
    print("reference program: rc", r.returncode)
    print((r.stdout + r.stderr)[-1500:])
    f = os.path.join(d, "MONORTM.OUT")
    if os.path.exists(f):
        print(open(f).read()[:3000])


if __name__ == "__main__":
    main()
