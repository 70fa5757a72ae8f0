#!/usr/bin/env python3
"""One seed of the GPU fuzz in detail: where the largest deviation from the oracle sits (layer, molecule, wavenumber) and the
state of that layer.    python tools/fuzz_one.py SEED [SEED ...]      (GPU box, repo root)"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_fuzz_gpu as fz  # noqa: E402
from monortm_amd import api  # noqa: E402
from oracle.pyoracle import Oracle  # noqa: E402

w = tempfile.mkdtemp()
for seed in [int(x) for x in sys.argv[1:]]:
    t3, pr = fz.random_case(seed, w)
    exp = Oracle(t3, pr.wn[0], pr.wn[-1]).run(pr)
    rt = api.MonoRTM(t3, pr.wn[0], pr.wn[-1])
    got = rt.run([pr])[0]
    rt.close()
    print("seed", seed, "nwn", pr.nwn, "nlay", pr.nlay, "wn", pr.wn[0], "..", pr.wn[-1], "irt", pr.irt, "ibrd", pr.ibrd)
    for f in ("o", "o_by_mol", "oc", "o_clw", "rad", "tb"):
        a, b = np.asarray(getattr(exp, f), dtype=float), np.asarray(getattr(got, f), dtype=float)
        ok = np.isfinite(a) & np.isfinite(b) & (a != 0)
        if not ok.any():
            continue
        rel = np.zeros_like(a)
        rel[ok] = np.abs(b[ok] - a[ok]) / np.abs(a[ok])
        i = np.unravel_index(np.argmax(rel), rel.shape)
        print(f"  {f:9s} worst {rel[i]:.3e} at {tuple(int(x) for x in i)} exp {a[i]:.17g} got {b[i]:.17g}")
    a, b = np.asarray(exp.o_by_mol, dtype=float), np.asarray(got.o_by_mol, dtype=float)
    ok = np.isfinite(a) & (a != 0)
    rel = np.zeros_like(a)
    rel[ok] = np.abs(b[ok] - a[ok]) / np.abs(a[ok])
    lay, mol, iw = (int(x) for x in np.unravel_index(np.argmax(rel), rel.shape))
    print("  layer", lay, "P", pr.p[lay], "T", pr.t[lay], "molecule", mol + 1, "wn", pr.wn[iw], "column", np.asarray(pr.wkl)[lay][mol] if hasattr(pr, "wkl") else None)
    print("  per-molecule values at that (layer, wn): exp", a[lay, :, iw], "\n   got", b[lay, :, iw])
