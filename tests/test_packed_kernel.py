"""The layer-packed line-sum kernel (monortm_amd/csrc/lines_packed_kernel.hip: four-wave workgroups whose lanes are the
(layer, wavenumber) pairs of several layers of a profile) is an opt-in alternative to lines_kernel for channel sets that leave a
64-lane tile partly empty (MONORTM_LINES_KERNEL=p).  It must give the reference's results on everything lines_kernel is
held to: every golden fixture with at most 64 wavenumbers (1e-6 of the compiled reference), the single-precision fixture,
ragged batches against the oracle (layer counts that are not multiples of the packing, species broadening, Voigt shapes and
line coupling), bitwise determinism - and agree with lines_kernel to the last bits."""
import numpy as np
import pytest

from common import RTOL, Golden, compare, golden_names
from monortm_amd import api, synth, tape3

pytestmark = pytest.mark.gpu


@pytest.fixture()
def packed_kernel(monkeypatch):
    import torch

    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests need the MI355X")
    monkeypatch.setenv("MONORTM_LINES_KERNEL", "p")   # read by the library when a context is created
    yield
    monkeypatch.delenv("MONORTM_LINES_KERNEL", raising=False)


@pytest.mark.parametrize("name", golden_names())
def test_packed_kernel_matches_reference_golden(name, workdir, packed_kernel):
    g = Golden(name, workdir)
    if g.profiles[0].nwn > 64:
        pytest.skip("more than 64 wavenumbers: lines_kernel's multi-wave tiles serve this input")
    rt = api.MonoRTM(g.tape3, g.profiles[0].wn[0], g.profiles[0].wn[-1])
    for i, (pr, exp) in enumerate(zip(g.profiles, g.expected)):
        compare(rt.run([pr])[0], exp, rtol=RTOL, what=f"packed kernel {name}[{i}]")
    rt.close()


def test_packed_kernel_single_precision(workdir, packed_kernel):
    g = Golden("sgl_cloud_updown", workdir)
    if g.profiles[0].nwn > 64:
        pytest.skip("more than 64 wavenumbers")
    rt = api.MonoRTM(g.tape3, g.profiles[0].wn[0], g.profiles[0].wn[-1], real_kind=4)
    for i, (pr, exp) in enumerate(zip(g.profiles, g.expected)):
        compare(rt.run([pr])[0], exp, rtol=2e-4, what=f"packed kernel sgl[{i}]", rad_floor=1e-30)
    rt.close()


@pytest.mark.parametrize("nwn,ibrd,real_kind", [(50, 0, 8), (64, 0, 8), (37, 1, 8), (50, 0, 4), (9, 0, 8)])
def test_packed_kernel_ragged_batch(workdir, packed_kernel, monkeypatch, nwn, ibrd, real_kind):
    """Ragged batches (layer counts that are no multiples of the layers per workgroup, cloud, both geometries; a line list with
    speed-dependent Voigt shapes, line coupling and species-broadening data) against the oracle; lines_kernel on the same inputs
    agrees to 1e-11 (the same terms; a line may take the tested loop in one kernel and the untested one in the other)."""
    from oracle.pyoracle import Oracle

    rec = synth.synthetic_lines(700, seed=77 + nwn, sdep_frac=0.2, lc_frac=0.5)
    if ibrd:  # species-by-species broadening data on a random subset of the physical lines
        rng = np.random.default_rng(11)
        n = len(rec.vnu)
        phys = rec.iflg >= 0
        dat = np.zeros((n, 21), np.float32)
        dat[:, 0::3] = rng.uniform(0.03, 0.15, (n, 7))
        dat[:, 1::3] = rng.uniform(0.4, 0.8, (n, 7))
        dat[:, 2::3] = rng.uniform(-0.004, 0.004, (n, 7))
        rec.brd_flg = (rng.random((n, 7)) < 0.3).astype(np.int32) * phys[:, None]
        rec.brd_dat = dat * phys[:, None]
    t3 = f"{workdir}/TAPE3_packed_{nwn}_{ibrd}_{real_kind}"
    tape3.write_tape3(t3, rec)
    wn = synth.c2_channels(nwn, seed=5)
    profs = [synth.perturbed_profile(300 + i, wn, nlay=nl, cloud=(i % 2 == 0), irt=(1 if i % 3 == 0 else 3))
             for i, nl in enumerate((64, 40, 17, 64, 33, 5, 64, 1))]
    for pr in profs:
        pr.ibrd = ibrd
    rt = api.MonoRTM(t3, wn[0], wn[-1], real_kind=real_kind)
    got = rt.run(profs)
    again = rt.run(profs)
    orc = Oracle(t3, wn[0], wn[-1])
    tol = RTOL if real_kind == 8 else 1e-3   # (float sums of several hundred terms of either sign; lines_kernel is held to the same below)
    for i, pr in enumerate(profs):
        compare(got[i], orc.run(pr), rtol=tol, what=f"packed kernel ragged[{i}] nlay={pr.nlay}", rad_floor=1e-30 if real_kind == 4 else 0.0)
        assert np.array_equal(got[i].o_by_mol, again[i].o_by_mol)      # deterministic
    rt.set_option("lines_kernel", "wn")
    ref = rt.run(profs)
    for i in range(len(profs)):
        compare(got[i], ref[i], rtol=1e-11 if real_kind == 8 else 2e-5, what=f"packed vs default kernel [{i}]",
                rad_floor=1e-30 if real_kind == 4 else 0.0)
    rt.close()


def test_packed_kernel_temperature_stop(workdir, packed_kernel):
    """A layer outside 70-3000 K stops the reference (tips_2003.f90:277) whatever the line window holds."""
    rec = synth.synthetic_lines(60, seed=3)
    t3 = f"{workdir}/TAPE3_packed_t"
    tape3.write_tape3(t3, rec)
    wn = synth.c2_channels(50, seed=5)
    pr = synth.perturbed_profile(1, wn, nlay=12)
    pr.t = pr.t.copy()
    pr.t[7] = 55.0
    rt = api.MonoRTM(t3, wn[0], wn[-1])
    with pytest.raises(api.MonoRTMError) as e:
        rt.run([pr])
    assert e.value.code == 4   # MONORTM_ETEMP
    rt.close()
