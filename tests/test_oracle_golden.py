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
