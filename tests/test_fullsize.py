"""BASELINE.json configurations at FULL size on the GPU.  The CPU oracle cannot redo them entirely in seconds, so
each case combines (i) an oracle comparison on a sample of the work and (ii) size-independent properties:
determinism (bitwise), batch == single, additivity of the line sum over a split line list, physical bounds."""
import numpy as np
import pytest

from common import RTOL, compare
from monortm_amd import api, synth, tape3
from monortm_amd.tape3 import LineRecords

pytestmark = pytest.mark.gpu


def _subset(rec: LineRecords, sel) -> LineRecords:
    return LineRecords(**{k: getattr(rec, k)[sel] for k in ("vnu", "sp", "alfa", "epp", "mol", "hwhm", "tmpalf", "pshift", "iflg",
                                                             "brd_flg", "brd_dat", "sdep")})


def test_c4_shard_full_size(workdir):
    """configs[3] per-GPU share: 128 profiles x 64 layers x 50 channels x 500 lines (the bench workload)."""
    from oracle.pyoracle import Oracle

    rec = synth.synthetic_lines(500)
    t3 = f"{workdir}/TAPE3_c4"
    tape3.write_tape3(t3, rec)
    wn = synth.c2_channels(50)
    profs = [synth.perturbed_profile(i, wn, nlay=64) for i in range(128)]
    rt = api.MonoRTM(t3, wn[0], wn[-1])
    b = api.DeviceBatch(rt, profs)
    b.step()
    b.check()
    first = [x.clone() for x in (b.O, b.OBM, b.TB, b.RAD)]
    b.step()
    b.check()
    for x, y in zip(first, (b.O, b.OBM, b.TB, b.RAD)):
        assert bool((x == y).all()), "two passes over the same resident batch must agree bitwise"
    tb = b.TB.cpu().numpy()
    assert np.all(np.isfinite(tb)) and tb.min() > 2.7 and tb.max() < 320.0
    orc = Oracle(t3, wn[0], wn[-1])
    dumps = b.dumps(profs)
    for i in (0, 77, 127):
        compare(dumps[i], orc.run(profs[i]), rtol=RTOL, what=f"c4 profile {i}")
        single = rt.run([profs[i]])[0]      # different launch geometry (line slicing): same sums to rounding
        assert np.allclose(single.o, dumps[i].o, rtol=1e-12, atol=0)
    rt.close()


def test_c3_full_size_sampled_and_additive(workdir):
    """configs[2]: 1 profile x 64 layers x 10000-wavenumber grid x 100000 lines = 6.4e10 evaluations.
    Oracle on 64 wavenumbers of the grid (4e8 evaluations); additivity: the per-molecule optical depths of the
    full list equal the sum over the two halves of the list (line sum is linear in the line set)."""
    from oracle.pyoracle import Oracle

    rec = synth.synthetic_lines(100000, seed=20261004)
    t3 = f"{workdir}/TAPE3_c3"
    tape3.write_tape3(t3, rec)
    a = synth.standard_atmosphere(64)
    wn = 0.5 + 0.005 * np.arange(10000)
    mk = lambda w, dv: synth.Profile(wn=w, p=a["p"], t=a["t"], tz=a["tz"], wkl=a["wkl"], wbrodl=a["wbrodl"], clw=a["clw"],  # noqa: E731
                                    irt=3, dvset=dv)
    rt = api.MonoRTM(t3, wn[0], wn[-1])
    assert rt.line_count(0) == 100000
    O, OBM, OC, OCLW = rt.modm([mk(wn, 0.005)])
    # 64 grid points x 64 layers x 100000 lines = 4e8 oracle evaluations: both ends of the grid, points either side of tile,
    # pair, four and eight boundaries of the 128-point tiles (round 5's levels of far_kernel), and 50 seeded random points
    rng = np.random.default_rng(64)
    idx = np.unique(np.concatenate([[0, 1, 127, 128, 255, 256, 511, 512, 1023, 1024, 4999, 5000, 8191, 9983, 9984, 9999],
                                    rng.choice(10000, 50, replace=False)]))[:64]
    orc = Oracle(t3, wn[0], wn[-1])
    ref = orc.run(mk(wn[idx], 0.0))
    got = OBM[0][:, :, idx]
    scale = np.maximum(np.abs(ref.o_by_mol), 1e-6 * np.abs(ref.o)[:, None, :])
    assert np.max(np.abs(got - ref.o_by_mol) / scale) < RTOL
    rt.close()
    # additivity over a split of the list (odd / even records), same wavenumber sample
    parts = []
    for par in (0, 1):
        sub = _subset(rec, np.arange(len(rec)) % 2 == par)
        p = f"{workdir}/TAPE3_c3_{par}"
        tape3.write_tape3(p, sub)
        r = api.MonoRTM(p, wn[0], wn[-1])
        parts.append(r.modm([mk(wn[idx], 0.0)])[1][0])
        r.close()
    tot = parts[0] + parts[1]
    assert np.max(np.abs(tot - got) / np.maximum(np.abs(got), 1e-300)) < 1e-9


def test_c3_whole_tile_against_oracle(workdir):
    """configs[2], whole stretches of the grid, every per-molecule optical depth compared with the oracle (4 layers spread over
    the column x 100000 lines, ~1.6e8 oracle evaluations per stretch).  The stretches follow round 5's structure - 79 one-wave
    tiles of 128 wavenumbers (the last one a stub of 16), far_kernel's levels of tiles, pairs, fours, eights:
      (i)   grid points 896 .. 1407 = tiles 7-10: crosses the eight-tile boundary at 1024 (and a four and two pair boundaries);
      (ii)  the last 144 points = tile 77 and the 16-point stub 78 (the stub fails the plan's Voigt guard and INHERITS the far lines
            of its nearest passing ancestor);
      (iii) the first 128 points (0.5-1.135 cm-1: lines whose negative resonance is within reach, two-resonance classes)."""
    from oracle.pyoracle import Oracle

    rec = synth.synthetic_lines(100000, seed=20261004)
    t3 = f"{workdir}/TAPE3_c3"
    tape3.write_tape3(t3, rec)
    a = synth.standard_atmosphere(64)
    wn = 0.5 + 0.005 * np.arange(10000)
    full = synth.Profile(wn=wn, p=a["p"], t=a["t"], tz=a["tz"], wkl=a["wkl"], wbrodl=a["wbrodl"], clw=a["clw"], irt=3, dvset=0.005)
    rt = api.MonoRTM(t3, wn[0], wn[-1])
    OBM = rt.modm([full])[1][0]          # [nlay, nmol, nwn]
    rt.close()
    lay = np.array([0, 21, 44, 63])
    tz4 = np.concatenate([a["tz"][lay], a["tz"][lay[-1] + 1:lay[-1] + 2]])
    orc = Oracle(t3, wn[0], wn[-1])   # same TAPE3 window as the GPU context
    for what, sl in (("tiles 7-10 across the eight-tile boundary", slice(896, 1408)), ("tile 77 and the stub tile", slice(10000 - 144, 10000)),
                     ("the first tile", slice(0, 128))):
        sub = synth.Profile(wn=wn[sl], p=a["p"][lay], t=a["t"][lay], tz=tz4, wkl=a["wkl"][lay], wbrodl=a["wbrodl"][lay],
                            clw=a["clw"][lay], irt=3, dvset=0.0)
        ref = orc.run(sub)
        got = OBM[lay][:, :, sl]
        scale = np.maximum(np.abs(ref.o_by_mol), 1e-6 * np.abs(ref.o)[:, None, :])
        err = np.abs(got - ref.o_by_mol) / scale
        assert err.max() < RTOL, (what, err.max(), np.unravel_index(np.argmax(err), err.shape))
        assert err.max() < 1e-9, (what, err.max())   # observed ~1e-13: the far-field series and the regrouped Lorentz sums are far inside 1e-6
    orc.close()


def test_c4_full_batch_on_one_gpu(workdir):
    """configs[3] WHOLE: 1024 profiles x 64 layers x 50 channels resident on one device (184 MB of per-molecule optical
    depths) - batch indexing at full size.  The whole batch takes lines_ms_kernel (round 6: six states a wave, the class of a
    line is the most general over the states of its wave, four lines share a reciprocal), the last shard of 128 profiles on its
    own takes lines_kernel: the two agree to the rounding of the regrouped sums (1e-12 of a cell's optical depth); with
    lines_kernel forced for both (one state per wave: a profile's arithmetic does not depend on its neighbours) the shard
    reproduces its slice of the big batch bit for bit.  Two profiles are checked against the oracle."""
    from oracle.pyoracle import Oracle

    rec = synth.synthetic_lines(500)
    t3 = f"{workdir}/TAPE3_c4"
    tape3.write_tape3(t3, rec)
    wn = synth.c2_channels(50)
    profs = [synth.perturbed_profile(i, wn, nlay=64) for i in range(1024)]
    rt = api.MonoRTM(t3, wn[0], wn[-1])
    rt.set_option("lines_kernel", "wn")
    b = api.DeviceBatch(rt, profs)
    b.step()
    b.check()
    wn_o, wn_obm = b.O[896:].cpu().numpy(), b.OBM[896:].cpu().numpy()
    rt.set_option("lines_kernel", "auto")
    b.step()
    b.check()
    tb = b.TB.cpu().numpy()
    assert tb.shape == (1024, 50) and np.all(np.isfinite(tb)) and tb.min() > 2.7 and tb.max() < 320.0
    big_o, big_obm, big_rad = b.O[896:].cpu().numpy(), b.OBM[896:].cpu().numpy(), b.RAD[896:].cpu().numpy()
    orc = Oracle(t3, wn[0], wn[-1])
    for i in (0, 1023):
        one = api.DeviceBatch(rt, [profs[i]])
        one.step()
        compare(one.dumps([profs[i]])[0], orc.run(profs[i]), rtol=RTOL, what=f"c4 profile {i}")
        assert np.allclose(one.TB.cpu().numpy()[0], tb[i], rtol=1e-12, atol=0)
    del b
    s = api.DeviceBatch(rt, profs[896:])
    s.step()
    s.check()
    so, sobm = s.O.cpu().numpy(), s.OBM.cpu().numpy()
    assert np.array_equal(so, wn_o) and np.array_equal(sobm, wn_obm)          # lines_kernel: bit for bit
    assert not np.array_equal(sobm, big_obm)                                   # (the big batch did take the other kernel)
    assert np.max(np.abs(so - big_o) / np.maximum(np.abs(big_o), 1e-300)) < 1e-12
    assert np.max(np.abs(sobm - big_obm) / np.maximum(np.abs(big_o)[:, :, None, :], 1e-300)) < 1e-12
    # the radiance recurrence splits the layers into 16 groups for <= 255 workgroups and 8 otherwise: same sums to rounding
    assert np.allclose(s.RAD.cpu().numpy(), big_rad, rtol=1e-13, atol=0)
    rt.close()


def test_c5_full_batch_single_precision(workdir):
    """configs[4] WHOLE: 256 profiles x 200 channels x 64 layers, up- and downwelling with liquid cloud, real_kind 4, on
    one device; every 17th profile against the double-precision context (5e-5: float Lorentz loop), three against the
    oracle, and the host-buffer route must equal the resident route bitwise."""
    from oracle.pyoracle import Oracle

    rec = synth.synthetic_lines(500)
    t3 = f"{workdir}/TAPE3_c5f"
    tape3.write_tape3(t3, rec)
    wn = synth.c2_channels(200)
    profs = [synth.perturbed_profile(i, wn, nlay=64, cloud=True, irt=(1 if i % 2 == 0 else 3)) for i in range(256)]
    rt4 = api.MonoRTM(t3, wn[0], wn[-1], real_kind=4)
    b = api.DeviceBatch(rt4, profs)
    b.step()
    b.check()
    d4 = b.dumps(profs)
    host = rt4.run(profs)   # the same batch (= the same kernel configuration) through host buffers
    for x, y in zip(host[200:256], d4[200:256]):
        assert np.array_equal(x.o, y.o) and np.array_equal(x.tb, y.tb) and np.array_equal(x.o_by_mol, y.o_by_mol)
    # a batch below 8192 (profile, layer) states takes two tiles of two wavenumbers per lane instead of one of four (api.hip
    # lines_config): the radiation term and the molecule sums are held in double there - the same results to float rounding
    part = rt4.run(profs[200:256])
    for x, y in zip(part, d4[200:256]):
        compare(x, y, rtol=2e-6, what="c5 two-wavenumber tiles vs four-wavenumber tile", rad_floor=1e-30)
    rt4.close()
    rt8 = api.MonoRTM(t3, wn[0], wn[-1])
    sel = list(range(0, 256, 17))
    d8 = rt8.run([profs[i] for i in sel])
    for k, i in enumerate(sel):
        compare(d4[i], d8[k], rtol=5e-5, what=f"c5 full batch real4 vs real8 profile {i}", rad_floor=1e-30)
    rt8.close()
    orc = Oracle(t3, wn[0], wn[-1])
    for i in (0, 129, 255):
        compare(d4[i], orc.run(profs[i]), rtol=5e-5, what=f"c5 full batch profile {i}", rad_floor=1e-30)


def test_c5full_exact_bench_workload(workdir):
    """EXACTLY the workload bench.py times as configs[4]: bench.build_workload("c5full") - 256 cloudy profiles
    x both views (512 runs) x 64 layers x 200 channels U(0.3, 6.5) cm-1 x 500 lines, real_kind 4: the one workload that takes
    lines_kernel<float,1,4> WITH the FULL class.  Oracle on four runs (up- and down-views, first and last profile and two in
    between), bitwise determinism of a second step, and the host-buffer route equals the resident route bitwise."""
    import sys

    from common import ROOT
    from oracle.pyoracle import Oracle

    sys.path.insert(0, ROOT)
    import bench

    rec, profs, desc, real_kind, _ = bench.build_workload("c5full", 0, 128, 1)
    assert real_kind == 4 and len(profs) == 512 and profs[0].nwn == 200 and profs[0].wn[-1] <= 6.5
    assert {p.irt for p in profs} == {1, 3} and all(p.clw.max() > 0 for p in profs[:8])
    t3 = f"{workdir}/TAPE3_c5full"
    tape3.write_tape3(t3, rec)
    wn = profs[0].wn
    rt4 = api.MonoRTM(t3, wn[0], wn[-1], real_kind=4)
    b = api.DeviceBatch(rt4, profs)
    b.step()
    b.check()
    d4 = b.dumps(profs)
    first = {k: getattr(b, k).clone() for k in ("O", "OBM", "OC", "RAD", "TB", "TMR")}
    b.step()
    b.check()
    for k, v in first.items():
        assert torch_equal(getattr(b, k), v), f"c5full: {k} differs between two steps of the same batch"
    orc = Oracle(t3, wn[0], wn[-1])
    sel = (0, 1, 254, 511)
    assert {profs[i].irt for i in sel} == {1, 3}
    for i in sel:
        compare(d4[i], orc.run(profs[i]), rtol=5e-5, what=f"c5full run {i} irt={profs[i].irt}", rad_floor=1e-30)
    host = rt4.run(profs)   # the same batch (= the same kernel configuration) through host buffers
    for i in list(range(0, 512, 37)) + [511]:
        x, y = host[i], d4[i]
        assert np.array_equal(x.o, y.o) and np.array_equal(x.tb, y.tb) and np.array_equal(x.o_by_mol, y.o_by_mol) and np.array_equal(x.rad, y.rad), i
    rt4.close()


def test_c2real_bench_workload_against_oracle(workdir):
    """bench.py's c2real workload (the real-file-like line list of tests/golden/real_like.npz through 40 sounder channels, 32
    profiles x 64 layers to 60 km): first, middle and last profile against the oracle, whole batch deterministic."""
    import sys

    from common import ROOT
    from oracle.pyoracle import Oracle

    sys.path.insert(0, ROOT)
    import bench

    rec, profs, desc, real_kind, t3kw = bench.build_workload("c2real", 0, 128, 1)
    assert real_kind == 8 and len(profs) == 32 and t3kw.get("second_header")
    t3 = f"{workdir}/TAPE3_c2real"
    tape3.write_tape3(t3, rec, **t3kw)
    wn = profs[0].wn
    rt = api.MonoRTM(t3, wn[0], wn[-1])
    b = api.DeviceBatch(rt, profs)
    b.step()
    b.check()
    d = b.dumps(profs)
    first = b.OBM.clone()
    b.step()
    assert torch_equal(b.OBM, first)
    orc = Oracle(t3, wn[0], wn[-1])
    for i in (0, 15, 31):
        compare(d[i], orc.run(profs[i]), rtol=RTOL, what=f"c2real profile {i}")
    rt.close()


def torch_equal(a, b):
    import torch

    return bool(torch.equal(a, b))


def test_c5_shape_cloud_up_and_down(workdir):
    """configs[4] flavour on one GPU: 32 profiles (256 / 8) x 200 channels (0.3-6.5 cm-1) x 64 layers with liquid cloud,
    downwelling and upwelling; oracle on three of them."""
    from oracle.pyoracle import Oracle

    rec = synth.synthetic_lines(500)
    t3 = f"{workdir}/TAPE3_c5"
    tape3.write_tape3(t3, rec)
    wn = np.sort(np.random.default_rng(55).uniform(0.3, 6.5, 200))
    profs = [synth.perturbed_profile(i, wn, nlay=64, cloud=True, irt=(3 if i % 2 == 0 else 1)) for i in range(32)]
    rt = api.MonoRTM(t3, wn[0], wn[-1])
    dumps = rt.run(profs)
    orc = Oracle(t3, wn[0], wn[-1])
    for i in (0, 13, 31):
        assert profs[i].clw.max() > 0
        compare(dumps[i], orc.run(profs[i]), rtol=RTOL, what=f"c5 profile {i} irt={profs[i].irt}")
    rt.close()
    # configs[4] is a single-precision configuration: the same batch through the real_kind = 4 context
    rt4 = api.MonoRTM(t3, wn[0], wn[-1], real_kind=4)
    d4 = rt4.run(profs)
    for i in (0, 13, 31):
        compare(d4[i], orc.run(profs[i]), rtol=5e-5, what=f"c5 real4 profile {i}", rad_floor=1e-30)
    for i in range(32):  # and against the double-precision GPU results for every profile
        compare(d4[i], dumps[i], rtol=5e-5, what=f"c5 real4 vs real8 profile {i}", rad_floor=1e-30)
    rt4.close()
