"""lines_ms_kernel (round 6: several atmospheric states per wave, five wavenumbers per lane - monortm_amd/csrc/lines_ms_kernel.hip)
against the reference's fixtures, the oracle and lines_kernel.  The kernel is chosen automatically for large batches of states on
sparse channel sets (configs[3]); here the option `lines_kernel = ms` forces it onto every case its layout can take (double
precision, <= 64 wavenumbers), including single profiles, ragged batches, coupled O2, Voigt candidates, species broadening,
NaN / negative amplitudes and the temperature stop.  Tolerance: north_star's 1e-6 against the reference; against lines_kernel the
two agree to the rounding of the shared reciprocals."""
import numpy as np
import pytest

from common import RTOL, Golden, compare, compare_nan_aware, golden_names, per_molecule_errors
from monortm_amd import api, synth, tape3

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests need the MI355X")
    api.load_library()
    return True


def _rt(t3, wn, kernel):
    rt = api.MonoRTM(t3, wn[0], wn[-1])
    rt.set_option("lines_kernel", kernel)
    return rt


def _small(name, workdir):
    g = Golden(name, workdir)
    return g if g.profiles[0].nwn <= 64 else None


@pytest.mark.parametrize("name", golden_names())
def test_ms_matches_reference_golden(name, workdir, gpu):
    g = _small(name, workdir)
    if g is None:
        pytest.skip("more than 64 wavenumbers: not a shape of lines_ms_kernel")
    rt = _rt(g.tape3, g.profiles[0].wn, "ms")
    for i, (pr, exp) in enumerate(zip(g.profiles, g.expected)):
        got = rt.run([pr])[0]
        compare(got, exp, rtol=RTOL, what=f"ms {name}[{i}]")
        pm = per_molecule_errors(got, exp)   # every molecule by itself, without compare()'s floor
        assert not (pm > RTOL).any(), f"ms {name}[{i}]: per-molecule errors {pm}"
    # ... and the whole fixture as ONE batch where its profiles share the scalar options (several states per wave)
    if len(g.profiles) > 1 and len({(p.nwn, p.nmol, p.ibrd, p.sclcpl, p.sclhw, p.y0res, p.dvset) for p in g.profiles}) == 1 and \
            all(np.array_equal(p.wn, g.profiles[0].wn) and np.array_equal(p.cntnm, g.profiles[0].cntnm) for p in g.profiles):
        for i, (got, exp) in enumerate(zip(rt.run(g.profiles), g.expected)):
            compare(got, exp, rtol=RTOL, what=f"ms batch {name}[{i}]")
    rt.close()


def test_forced_ms_is_not_a_silent_fallback(workdir, gpu):
    """A single profile under `lines_kernel = ms` really takes lines_ms_kernel (one state per wave then; its sums differ from
    lines_kernel's in the last bits) - the golden tests above would pass on lines_kernel too."""
    g = Golden("c2_base", workdir)
    a, b = _both(g.tape3, g.profiles[:1])
    assert not np.array_equal(a[0].o_by_mol, b[0].o_by_mol)
    _close(b[0], a[0], "c2_base")


def test_ms_nan_column_matches_reference(workdir, gpu):
    g = Golden("nan_column", workdir)
    n = 0
    for i, (pr, exp) in enumerate(zip(g.profiles, g.expected)):
        if pr.nwn > 64:
            continue
        rt = _rt(g.tape3, pr.wn, "ms")
        compare_nan_aware(rt.run([pr])[0], exp, rtol=RTOL, what=f"ms nan_column[{i}]")
        rt.close()
        n += 1
    assert n >= 1


def _both(t3, profs):
    out = {}
    for k in ("wn", "ms"):
        rt = _rt(t3, profs[0].wn, k)
        out[k] = rt.run(profs)
        rt.close()
    return out["wn"], out["ms"]


def _close(a, b, what, tol=1e-11):
    for f in ("o", "o_by_mol", "rad", "tb", "tmr", "trtot", "rup", "rdn"):
        x, y = np.asarray(getattr(a, f)), np.asarray(getattr(b, f))
        scale = np.maximum(np.abs(y), 1e-9 * np.abs(y).max() if y.size else 1.0)
        if f == "o_by_mol":
            scale = np.maximum(np.abs(y), 1e-9 * np.abs(np.asarray(b.o))[:, None, :])
        err = float(np.max(np.abs(x - y) / np.maximum(scale, 1e-300))) if x.size else 0.0
        assert err <= tol, f"{what}: {f} differs by {err:.2e} between lines_kernel and lines_ms_kernel"


@pytest.mark.parametrize("nwn,nprof", [(50, 13), (40, 8), (64, 5), (33, 20), (7, 3), (1, 2)])
def test_ms_equals_wn_kernel_ragged_batches(workdir, gpu, nwn, nprof):
    """Ragged layer counts, cloud, both geometries, channel counts that fill 10 / 8 / 13 / 7 / 2 / 1 lanes per state, profile counts
    that leave the last group of a layer short; coupled and speed-dependent lines in the list."""
    from oracle.pyoracle import Oracle

    rec = synth.synthetic_lines(420, seed=600 + nwn, sdep_frac=0.15, lc_frac=0.4)
    t3 = f"{workdir}/TAPE3_ms_{nwn}"
    tape3.write_tape3(t3, rec)
    wn = synth.c2_channels(nwn, seed=nwn)
    lays = [64, 40, 17, 64, 33, 5, 64, 64, 12, 50, 64, 3, 64]
    profs = [synth.perturbed_profile(300 + i, wn, nlay=lays[i % len(lays)], cloud=(i % 2 == 0), irt=(1 if i % 3 == 0 else 3)) for i in range(nprof)]
    a, b = _both(t3, profs)
    orc = Oracle(t3, wn[0], wn[-1])
    for i, pr in enumerate(profs):
        _close(b[i], a[i], f"nwn={nwn} profile {i}")
        if i < 4:
            compare(b[i], orc.run(pr), rtol=RTOL, what=f"ms vs oracle nwn={nwn} [{i}]")
    orc.close()


def test_ms_voigt_and_coupling_live(workdir, gpu):
    """bench.py's c2lc shape: every O2 line first-order coupled, 10 % speed dependent, the model top at 0.004 hPa and channels on
    line centres - Voigt candidates in most of the upper states, different ones per state of a wave."""
    import bench

    rec, profs, _, _, _ = bench.build_workload("c2lc", 0, 14)
    t3 = f"{workdir}/TAPE3_ms_c2lc"
    tape3.write_tape3(t3, rec)
    a, b = _both(t3, profs)
    from oracle.pyoracle import Oracle

    orc = Oracle(t3, profs[0].wn[0], profs[0].wn[-1])
    orc.census(reset=True)
    for i, pr in enumerate(profs):
        _close(b[i], a[i], f"c2lc profile {i}", tol=1e-10)
        if i < 3:
            compare(b[i], orc.run(pr), rtol=RTOL, what=f"ms vs oracle c2lc [{i}]")
    assert orc.census()["voigt"] > 0
    orc.close()


def test_ms_species_broadening(workdir, gpu):
    import bench

    rec, profs, _, _, _ = bench.build_workload("c4brd", 0, 9)
    t3 = f"{workdir}/TAPE3_ms_brd"
    tape3.write_tape3(t3, rec)
    a, b = _both(t3, profs)
    for i in range(len(profs)):
        _close(b[i], a[i], f"c4brd profile {i}")


def test_ms_zero_columns_and_temperature_stop(workdir, gpu):
    """A molecule without a column in SOME states of a wave (the wave walks its lines for the others: the result must be exactly
    zero there, as the reference skips the molecule - src/modm.f90:318-321), and a layer at 50 K (TIPS stop, MONORTM_ETEMP)."""
    rec = synth.synthetic_lines(300, seed=77)
    t3 = f"{workdir}/TAPE3_ms_zero"
    tape3.write_tape3(t3, rec)
    wn = synth.c2_channels(50)
    profs = [synth.perturbed_profile(900 + i, wn, nlay=20) for i in range(7)]
    for i in (1, 4):
        profs[i].wkl[:, 2] = 0.0      # no O3 anywhere in two of the profiles
        profs[i].wkl[5:9, 0] = 0.0    # no H2O in four layers
    a, b = _both(t3, profs)
    for i in range(len(profs)):
        _close(b[i], a[i], f"zero-column profile {i}")
    for i in (1, 4):
        assert not b[i].o_by_mol[:, 2, :].any() and not b[i].o_by_mol[5:9, 0, :].any()
        assert b[i - 1].o_by_mol[:, 2, :].any()
    cold = [synth.perturbed_profile(950 + i, wn, nlay=20) for i in range(7)]
    cold[3].t[11] = 50.0
    rt = _rt(t3, wn, "ms")
    with pytest.raises(api.MonoRTMError) as e:
        rt.run(cold)
    assert e.value.code == 4   # MONORTM_ETEMP
    rt.close()


def test_ms_is_chosen_by_rounds_of_waves(workdir, gpu):
    """auto: lines_ms_kernel where the batch makes whole rounds of its waves (6 states a wave on 50 channels, 4096 wave slots):
    384 profiles x 64 layers = exactly one round takes it, 128 profiles (a third of a round) keep lines_kernel, and 512 (one round
    and a third) are SPLIT - the first 384 profiles through lines_ms_kernel, the other 128 through lines_kernel.  The forced kernels
    differ from each other in the last bits, so which one served a profile shows in its results."""
    rec = synth.synthetic_lines(120, seed=5)
    t3 = f"{workdir}/TAPE3_ms_auto"
    tape3.write_tape3(t3, rec)
    wn = synth.c2_channels(50)
    profs = [synth.perturbed_profile(i % 40, wn, nlay=64, cloud=(i % 7 == 0), irt=(1 if i % 2 else 3)) for i in range(512)]
    for n in (384, 128, 512):
        res = {}
        for k in ("wn", "ms", "auto"):
            rt = _rt(t3, wn, k)
            d = rt.run(profs[:n])
            res[k] = (np.stack([x.o_by_mol for x in d]), np.stack([x.o for x in d]), np.stack([x.tb for x in d]))
            rt.close()
        assert not np.array_equal(res["ms"][0], res["wn"][0])
        if n == 384:
            assert all(np.array_equal(a, b) for a, b in zip(res["auto"], res["ms"])), "384 profiles: auto is not lines_ms_kernel"
        elif n == 128:
            assert all(np.array_equal(a, b) for a, b in zip(res["auto"], res["wn"])), "128 profiles: auto is not lines_kernel"
        else:
            # (the ms run of 512 profiles groups them exactly as the first part of the split does: groups of six from profile 0)
            assert all(np.array_equal(a[:384], b[:384]) for a, b in zip(res["auto"], res["ms"])), "512 profiles: the first 384 are not lines_ms_kernel's"
            assert all(np.array_equal(a[384:], b[384:]) for a, b in zip(res["auto"], res["wn"])), "512 profiles: the last 128 are not lines_kernel's"
            assert not np.array_equal(res["auto"][0][:384], res["wn"][0][:384])


def test_ms_step_in_a_hip_graph(workdir, gpu):
    """A resident batch that takes lines_ms_kernel (384 profiles: one round of its waves), its step captured into a HIP graph after
    one warm step (the scratch of the rare shapes is allocated outside the capture) and replayed: bitwise the stream launches."""
    rec = synth.synthetic_lines(150, seed=8)
    t3 = f"{workdir}/TAPE3_ms_graph"
    tape3.write_tape3(t3, rec)
    wn = synth.c2_channels(50)
    profs = [synth.perturbed_profile(i % 64, wn, nlay=64, cloud=(i % 5 == 0)) for i in range(384)]
    rt = api.MonoRTM(t3, wn[0], wn[-1])
    b = api.DeviceBatch(rt, profs)
    b.step()
    b.check()
    ref = [x.clone() for x in (b.O, b.OBM, b.TB)]
    b.capture()
    for x in (b.O, b.OBM, b.TB):
        x.zero_()
    b.replay()
    b.check()
    for x, y in zip(ref, (b.O, b.OBM, b.TB)):
        assert bool((x == y).all())
    rt.set_option("lines_kernel", "wn")
    b.step()
    b.check()
    assert not bool((ref[1] == b.OBM).all())   # (the captured step did take lines_ms_kernel)
    rt.close()


@pytest.mark.parametrize("seed", range(10))
def test_ms_batch_fuzz_against_oracle(workdir, gpu, seed):
    """Seeded random batches through lines_ms_kernel with several DIFFERENT states per wave (the class of a line, its window and
    the slots it reaches are then formed over the states of the wave): line lists of 150-600 lines with coupled and
    speed-dependent ones, 11-64 channels, 5-17 profiles of ragged depth, some with a model top at 0.01 hPa (Voigt candidates
    that differ between the states of a wave) - every profile against the oracle."""
    from oracle.pyoracle import Oracle

    rng = np.random.default_rng(7000 + seed)
    rec = synth.synthetic_lines(int(rng.integers(150, 600)), seed=7100 + seed, sdep_frac=float(rng.uniform(0, 0.3)), lc_frac=float(rng.uniform(0, 0.6)))
    t3 = f"{workdir}/TAPE3_msfz_{seed}"
    tape3.write_tape3(t3, rec)
    nwn = int(rng.integers(11, 65))
    wn = synth.c2_channels(nwn, seed=int(rng.integers(1, 1000)), hi=float(rng.uniform(6.0, 40.0)))
    if seed % 3 == 0:   # a few channels on line centres
        phys = (rec.iflg >= 0) & (rec.vnu > wn[0]) & (rec.vnu < wn[-1])
        cent = rec.vnu[phys][:: max(1, int(phys.sum()) // 5)][:5]
        wn = np.sort(np.concatenate([wn[: nwn - len(cent)], cent + rng.uniform(-1e-4, 1e-4, len(cent))]))
    nprof = int(rng.integers(5, 18))
    profs = [synth.perturbed_profile(8000 + 50 * seed + i, wn, nlay=int(rng.integers(3, 65)), cloud=bool(rng.integers(0, 2)),
                                     irt=int(rng.choice([1, 3])), ztop_km=float(rng.choice([32.0, 60.0, 80.0]))) for i in range(nprof)]
    rt = _rt(t3, wn, "ms")
    got = rt.run(profs)
    rt.close()
    orc = Oracle(t3, wn[0], wn[-1])
    for i, pr in enumerate(profs):
        compare(got[i], orc.run(pr), rtol=RTOL, what=f"ms batch fuzz seed {seed} profile {i} (nwn {nwn}, nlay {pr.nlay})")
    orc.close()
