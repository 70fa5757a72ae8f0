"""The state-lane line-sum kernel (monortm_amd/csrc/lines_state_kernel.hip: lane = (profile, layer), wave = 8 wavenumbers) is
an opt-in alternative to lines_kernel (MONORTM_LINES_KERNEL=state).  It must give the reference's results on everything the
default kernel is held to: every double-precision golden fixture (1e-6 of the compiled reference), the single-precision
fixture, a ragged batch against the oracle, sliced line lists, and bitwise determinism."""
import os

import numpy as np
import pytest

from common import RTOL, Golden, compare, golden_names
from monortm_amd import api, synth, tape3

pytestmark = pytest.mark.gpu


@pytest.fixture()
def state_kernel(monkeypatch):
    import torch

    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests need the MI355X")
    monkeypatch.setenv("MONORTM_LINES_KERNEL", "state")   # read by the library when a context is created
    yield
    monkeypatch.delenv("MONORTM_LINES_KERNEL", raising=False)


@pytest.mark.parametrize("name", golden_names())
def test_state_kernel_matches_reference_golden(name, workdir, state_kernel):
    g = Golden(name, workdir)
    rt = api.MonoRTM(g.tape3, g.profiles[0].wn[0], g.profiles[0].wn[-1])
    for i, (pr, exp) in enumerate(zip(g.profiles, g.expected)):
        compare(rt.run([pr])[0], exp, rtol=RTOL, what=f"state kernel {name}[{i}]")
    rt.close()


def test_state_kernel_single_precision(workdir, state_kernel):
    g = Golden("sgl_cloud_updown", workdir)
    rt = api.MonoRTM(g.tape3, g.profiles[0].wn[0], g.profiles[0].wn[-1], real_kind=4)
    for i, (pr, exp) in enumerate(zip(g.profiles, g.expected)):
        compare(rt.run([pr])[0], exp, rtol=2e-4, what=f"state kernel sgl[{i}]", rad_floor=1e-30)
    rt.close()


@pytest.mark.parametrize("nslice", [1, 3])
def test_state_kernel_ragged_batch_equals_default_kernel(workdir, state_kernel, monkeypatch, nslice):
    """A ragged batch (different layer counts, cloud, both geometries) against the oracle, with and without sliced line lists;
    and the default kernel on the same inputs agrees to 1e-12 (different association of the same terms)."""
    from oracle.pyoracle import Oracle

    rec = synth.synthetic_lines(700, seed=77, sdep_frac=0.2, lc_frac=0.5)
    t3 = f"{workdir}/TAPE3_state_{nslice}"
    tape3.write_tape3(t3, rec)
    wn = synth.c2_channels(50, seed=5)
    profs = [synth.perturbed_profile(300 + i, wn, nlay=nl, cloud=(i % 2 == 0), irt=(1 if i % 3 == 0 else 3))
             for i, nl in enumerate((64, 40, 17, 64, 33, 5, 64, 64))]
    monkeypatch.setenv("MONORTM_NSLICE", str(nslice))
    rt = api.MonoRTM(t3, wn[0], wn[-1])
    got = rt.run(profs)
    again = rt.run(profs)
    orc = Oracle(t3, wn[0], wn[-1])
    for i, pr in enumerate(profs):
        compare(got[i], orc.run(pr), rtol=RTOL, what=f"state kernel ragged[{i}] nlay={pr.nlay}")
        assert np.array_equal(got[i].o_by_mol, again[i].o_by_mol)      # deterministic
    rt.set_option("lines_kernel", "wn")
    ref = rt.run(profs)
    for i in range(len(profs)):
        compare(got[i], ref[i], rtol=1e-11, what=f"state vs default kernel [{i}]")
    rt.close()
