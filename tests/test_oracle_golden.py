"""Pin the CPU restatement (oracle/monortm_oracle.c) against outputs of the reference itself.

The golden .npz files were produced by tests/golden/make_golden.py from
oracle/_ref/harness_ref_dbl = the reference compiled from /root/reference by amdflang.
Two independent implementations of IEEE double arithmetic in the same operation order agree
to a few ulp; the bound used here (1e-10) is four orders tighter than the product tolerance.
"""
import pytest

from common import Golden, compare, golden_names
from oracle.pyoracle import Oracle

ORACLE_RTOL = 1e-10


@pytest.mark.parametrize("name", golden_names())
def test_oracle_matches_reference(name, workdir):
    g = Golden(name, workdir)
    pr0 = g.profiles[0]
    orc = Oracle(g.tape3, pr0.wn[0], pr0.wn[-1])
    for i, (pr, exp) in enumerate(zip(g.profiles, g.expected)):
        got = orc.run(pr)
        compare(got, exp, rtol=ORACLE_RTOL, what=f"{name}[{i}]")
    orc.close()


def test_golden_set_is_complete():
    assert {"c2_base", "voigt_regions", "line_coupling", "cloud_updown", "ir_grid_nmol22", "cntnm_factors",
            "lc_o2_random", "ibrd_species_broadening"} <= set(golden_names())


def test_fixture_branch_census(workdir):
    """The fixtures must keep EXERCISING the branches (a fixture edit that silently drops, say, Humlicek region IV would
    leave every parity test green): counted by the oracle over all double-precision fixtures."""
    tot = {}
    irt, ibrd, cloud, dv = set(), set(), False, False
    for name in golden_names():
        g = Golden(name, workdir)
        orc = Oracle(g.tape3, g.profiles[0].wn[0], g.profiles[0].wn[-1])
        orc.census(reset=True)
        for pr in g.profiles:
            orc.run(pr)
            irt.add(pr.irt)
            ibrd.add(pr.ibrd)
            cloud |= bool((pr.clw > 0).any())
            dv |= pr.dvset != 0
        c = orc.census()
        orc.close()
        for k, v in c.items():
            tot[k] = [a + b for a, b in zip(tot.get(k, [0] * len(v)), v)] if isinstance(v, list) else tot.get(k, 0) + v
    assert all(n >= lo for n, lo in zip(tot["w4_region"], (1000, 100, 100, 30))), tot["w4_region"]       # Humlicek I-IV
    assert all(n >= lo for n, lo in zip(tot["sd_region"], (200, 30, 100, 8))), tot["sd_region"]          # SD_Humlicek I-IV
    assert tot["voigt"] >= 1000 and tot["lorentz"] >= 100000 and tot["cut_rejected"] >= 10000
    assert tot["coupled"] >= 10000 and tot["coupled_m3"] >= 100 and tot["coupled_m5"] >= 100 and tot["coupled_voigt"] >= 10
    assert irt == {1, 2, 3} and ibrd == {0, 1} and cloud and dv


HARNESS = None


def _harness():
    import os

    from common import ROOT

    p = os.path.join(ROOT, "oracle", "_ref", "harness_ref_dbl_fast")
    return p if os.path.exists(p) else None


@pytest.mark.parametrize("seed", range(9000, 9064))
def test_oracle_matches_reference_on_fuzz_cases(seed, workdir):
    """Where the compiled reference is present (oracle/_ref, built from /root/reference by `make -C oracle ref`), pin the
    restatement on the seeded fuzz cases of tests/test_fuzz_gpu.py too: the GPU test then rests on the reference itself.
    NaN positions (the reference's own, see test_fuzz_gpu.py) must coincide."""
    import subprocess

    import numpy as np

    from monortm_amd import caseio
    from test_fuzz_gpu import random_case

    h = _harness()
    if h is None:
        pytest.skip("oracle/_ref/harness_ref_dbl_fast not built")
    t3, pr = random_case(seed, workdir)
    case, out = f"{workdir}/fz{seed}.bin", f"{workdir}/fz{seed}.out"
    caseio.write_case(case, [pr])
    r = subprocess.run([h, case, t3, out], cwd=workdir, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:]
    exp = caseio.read_dump(out)[0]
    got = Oracle(t3, pr.wn[0], pr.wn[-1]).run(pr)
    bad = ~np.isfinite(exp.o).all(axis=0) | ~np.isfinite(exp.tb)
    assert np.array_equal(~np.isfinite(got.o).all(axis=0) | ~np.isfinite(got.tb), bad)
    if bad.any():
        keep = ~bad
        cut = lambda d: caseio.Dump(d.o[:, keep], d.o_by_mol[:, :, keep], d.oc[:, :, keep], d.o_clw[:, keep], d.rup[keep],  # noqa: E731
                                    d.rdn[keep], d.trtot[keep], d.rad[keep], d.tb[keep], d.tmr[keep], d.tmpsfc_out)
        got, exp = cut(got), cut(exp)
    compare(got, exp, rtol=ORACLE_RTOL, what=f"fuzz seed {seed} (oracle vs reference)")
