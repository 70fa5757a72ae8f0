#!/usr/bin/env python3
"""More seeds of the GPU fuzz than the test suite carries (tests/test_fuzz_gpu.py generators): HIP against the CPU oracle.
    python tools/fuzz_more.py FIRST_SEED COUNT [allmol | dense]  (GPU box, repo root; dense = grids of 513-3000 wavenumbers: far_kernel)
Prints the worst relative error per output field and every seed above 1e-6."""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import common  # noqa: E402
import test_fuzz_gpu as fz  # noqa: E402
from monortm_amd import api  # noqa: E402
from oracle.pyoracle import Oracle  # noqa: E402


def main():
    first, count = int(sys.argv[1]), int(sys.argv[2])
    mode = sys.argv[3] if len(sys.argv) > 3 else ""
    gen = fz.dense_case if mode == "dense" else (fz.random_case_allmol if mode else fz.random_case)
    worst, bad = {}, []
    with tempfile.TemporaryDirectory() as wd:
        for seed in range(first, first + count):
            t3, pr = gen(seed, wd)
            rt = api.MonoRTM(t3, pr.wn[0], pr.wn[-1])
            got = rt.run([pr])[0]
            rt.close()
            exp = Oracle(t3, pr.wn[0], pr.wn[-1]).run(pr)
            # the reference (and the oracle) return NaN for some inputs (tests/test_fuzz_gpu.py): the same wavenumbers must be
            # NaN here, the others are compared
            nanw = ~np.isfinite(exp.o).all(axis=0) | ~np.isfinite(exp.tb)
            if nanw.any():
                from monortm_amd.caseio import Dump
                if not np.array_equal(~np.isfinite(got.o).all(axis=0) | ~np.isfinite(got.tb), nanw):
                    bad.append((seed, "NaN pattern differs from the oracle's"))
                keep = ~nanw
                if not keep.any():
                    continue

                def cut(d):
                    return Dump(d.o[:, keep], d.o_by_mol[:, :, keep], d.oc[:, :, keep], d.o_clw[:, keep], d.rup[keep], d.rdn[keep],
                                d.trtot[keep], d.rad[keep], d.tb[keep], d.tmr[keep], d.tmpsfc_out)
                got, exp = cut(got), cut(exp)
            try:
                common.compare(got, exp, rtol=1e-6, what=f"seed {seed}")
            except AssertionError as e:
                bad.append((seed, str(e)[:300]))
            fin = np.isfinite(exp.o) & (np.abs(exp.o) > 0)
            err = float(np.max(np.abs(got.o[fin] - exp.o[fin]) / np.abs(exp.o[fin]))) if fin.any() else 0.0
            worst[seed] = err
            os.remove(t3)
    v = np.array(list(worst.values()))
    print(f"{count} seeds from {first}: total optical depth, worst relative error {v.max():.3e} (seed {max(worst, key=worst.get)}), "
          f"median {np.median(v):.3e}; failures at 1e-6: {len(bad)}")
    for s, m in bad:
        print(" ", s, m)


if __name__ == "__main__":
    main()
