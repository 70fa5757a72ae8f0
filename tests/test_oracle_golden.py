"""Pin the CPU restatement (oracle/monortm_oracle.c) against outputs of the reference itself.

The golden .npz files were produced by tests/golden/make_golden.py from
oracle/_ref/harness_ref_dbl = the reference compiled from /root/reference by amdflang.
Two independent implementations of IEEE double arithmetic in the same operation order agree
to a few ulp; the bound used here (1e-10) is four orders tighter than the product tolerance.
"""
import numpy as np
import pytest

from common import ALLMOL_SKIP, TIPS_ISONM, Golden, compare, compare_nan_aware, golden_names, per_molecule_errors
from oracle.pyoracle import Oracle

ORACLE_RTOL = 1e-10
# Lines of molecules > 7: the reference's HALFWHM_C reads rho_molec(mol) beyond the 7-element array (src/modm.f90:845).  With
# hwhm = alfa (as in the fixture) the term is alfa0i*(RHORAT - g) + alfa0i*g with g = whatever the stack holds: equal to
# alfa0i*RHORAT up to eps*|g|/RHORAT, observed <= 1.5e-7 in the thinnest layer (RHORAT 5e-5).  Those molecules are held to the
# product tolerance, molecules 1-7 of the same fixture to ORACLE_RTOL.
ORACLE_RTOL_OOB = 1e-6


@pytest.mark.parametrize("name", golden_names())
def test_oracle_matches_reference(name, workdir):
    g = Golden(name, workdir)
    pr0 = g.profiles[0]
    orc = Oracle(g.tape3, pr0.wn[0], pr0.wn[-1])
    oob = name == "all_molecules"
    for i, (pr, exp) in enumerate(zip(g.profiles, g.expected)):
        got = orc.run(pr)
        compare(got, exp, rtol=ORACLE_RTOL_OOB if oob else ORACLE_RTOL, what=f"{name}[{i}]")
        if oob:
            e = per_molecule_errors(got, exp)
            assert e[:7].max() <= ORACLE_RTOL, e[:7]
            assert np.nanmax(e) <= ORACLE_RTOL_OOB, e
    orc.close()


def test_oracle_matches_reference_nan_column(workdir):
    """A NaN column amount in one layer (fixture nan_column, outputs of the compiled reference): the NaN positions of every
    output field and the finite values elsewhere are the reference's (src/modm.f90:384,432 add the NaN term)."""
    g = Golden("nan_column", workdir)
    orc = Oracle(g.tape3, g.profiles[0].wn[0], g.profiles[0].wn[-1])
    for i, (pr, exp) in enumerate(zip(g.profiles, g.expected)):
        assert np.isnan(exp.o_by_mol).any() and np.isfinite(exp.o_by_mol).any()
        compare_nan_aware(orc.run(pr), exp, rtol=ORACLE_RTOL, what=f"nan_column[{i}]")
    orc.close()


def test_negative_strength_fixture_is_negative(workdir):
    """The fixture must keep what it is there for: per-molecule optical depths of BOTH signs (negative strengths added with
    their sign), on the sparse channels and on the dense grid."""
    g = Golden("negative_strength", workdir)
    for exp in g.expected:
        assert (exp.o_by_mol < 0).any() and (exp.o_by_mol > 0).any() and np.isfinite(exp.o_by_mol).all()


def test_all_molecules_fixture_visits_every_tips_slot(workdir):
    """no fixture touched a line of molecule 5, 6 or 8-39 nor an isotopologue > 2, and the oracle shares the
    product's table header - so a common-mode table or TIPS error could not show.  The all_molecules fixture (outputs of the
    compiled reference) must evaluate shapes of EVERY (molecule, isotopologue <= min(9, ISONM)) that TIPS_2003 fills, with a
    per-molecule optical depth large enough for the comparison to be a relative one.  Molecules 19 and 20 are out: the
    reference returns NaN for them (reads beyond rho_molec, src/modm.f90:845)."""
    g = Golden("all_molecules", workdir)
    orc = Oracle(g.tape3, g.profiles[0].wn[0], g.profiles[0].wn[-1])
    orc.iso_census(reset=True)
    for pr in g.profiles:
        orc.run(pr)
    seen = orc.iso_census()
    orc.close()
    for mol in range(1, 40):
        for iso in range(1, 10):
            want = mol not in ALLMOL_SKIP and iso <= min(9, TIPS_ISONM[mol - 1])
            assert (seen[mol - 1, iso - 1] > 0) == want, (mol, iso, int(seen[mol - 1, iso - 1]))
    exp = g.expected[1]
    assert np.isfinite(exp.o_by_mol).all()
    peak = exp.o_by_mol.max(axis=(0, 2))
    tot = exp.o.max()
    for mol in range(1, 40):
        if mol not in ALLMOL_SKIP:
            assert peak[mol - 1] > 1e-4 * tot, (mol, peak[mol - 1], tot)   # 100 x compare()'s floor; per_molecule_errors() is floor-free


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


@pytest.mark.parametrize("seed", range(9100, 9124))
def test_oracle_matches_reference_on_all_molecule_fuzz(seed, workdir):
    """The all-molecule fuzz cases of tests/test_fuzz_gpu.py (NMOL up to 39, every isotopologue slot) through the compiled
    reference.  A warm-up profile without trace columns goes first: the reference's first LINES call reads start-up stack
    garbage (NaN) behind rho_molec(7) for molecule 8 (src/modm.f90:845); later calls find a finite stale value there, which
    hwhm = alfa multiplies by zero.  Molecules 1-7 are held to 1e-10, the others to 1e-6 (cancellation against that value)."""
    import subprocess

    from monortm_amd import caseio, synth
    from test_fuzz_gpu import random_case_allmol

    h = _harness()
    if h is None:
        pytest.skip("oracle/_ref/harness_ref_dbl_fast not built")
    t3, pr = random_case_allmol(seed, workdir)
    wk0 = pr.wkl[:1].copy()
    wk0[:, 7:] = 0.0
    warm = synth.Profile(wn=np.array([pr.wn[0], pr.wn[-1]]), p=pr.p[:1], t=pr.t[:1], tz=pr.tz[:2], wkl=wk0, wbrodl=pr.wbrodl[:1],
                         clw=pr.clw[:1], irt=3)
    case, out = f"{workdir}/fm{seed}.bin", f"{workdir}/fm{seed}.out"
    caseio.write_case(case, [warm, pr])
    r = subprocess.run([h, case, t3, out], cwd=workdir, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:]
    exp = caseio.read_dump(out)[1]
    assert np.isfinite(exp.o_by_mol).all() and np.isfinite(exp.tb).all()
    orc = Oracle(t3, pr.wn[0], pr.wn[-1])
    got = orc.run(pr)
    orc.close()
    compare(got, exp, rtol=ORACLE_RTOL_OOB, what=f"all-molecule fuzz seed {seed} (oracle vs reference)")
    e = per_molecule_errors(got, exp)
    assert not (e[:7] > ORACLE_RTOL).any() and not (e > ORACLE_RTOL_OOB).any(), e


def test_fuzz_seeds_include_a_nan_case(workdir):
    """At least one seed of the GPU fuzz (tests/test_fuzz_gpu.py FUZZ_SEEDS) makes the reference's algorithm return NaN columns
    (Rayleigh term over the radiation term at 0 cm-1): the NaN-pattern comparison of test_fuzz_against_oracle is live."""
    import test_fuzz_gpu as fz
    from oracle.pyoracle import Oracle

    assert 50269 in fz.FUZZ_SEEDS
    t3, pr = fz.random_case(50269, workdir)
    exp = Oracle(t3, pr.wn[0], pr.wn[-1]).run(pr)
    bad = ~np.isfinite(exp.o).all(axis=0) | ~np.isfinite(exp.tb)
    assert bad.any(), "seed 50269 no longer produces NaN columns: pick another NaN seed for FUZZ_SEEDS"
