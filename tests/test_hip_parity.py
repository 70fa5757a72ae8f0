"""GPU parity: the HIP path (through the C ABI) against (i) the reference's own outputs (golden
fixtures) and (ii) the CPU oracle on fresh seeded inputs.  Tolerance = north_star's 1e-6 relative on
brightness temperature, radiance and layer optical depths."""
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


@pytest.mark.parametrize("name", golden_names())
def test_hip_matches_reference_golden(name, workdir, gpu):
    g = Golden(name, workdir)
    pr0 = g.profiles[0]
    rt = api.MonoRTM(g.tape3, pr0.wn[0], pr0.wn[-1])
    for i, (pr, exp) in enumerate(zip(g.profiles, g.expected)):
        got = rt.run([pr])[0]
        compare(got, exp, rtol=RTOL, what=f"{name}[{i}]")
        # ... and every molecule by itself, WITHOUT the floor of compare() (cells where the molecule's optical depth is >= 1e-6 of its
        # own peak, however small a share of the cell's total: VERDICT r5 weak 1b); observed <= 2.5e-8 (NO in all_molecules), 1e-12 elsewhere
        pm = per_molecule_errors(got, exp)
        assert not (pm > RTOL).any(), f"{name}[{i}]: per-molecule errors {pm}"
    rt.close()


def test_hip_nan_column_matches_reference(workdir, gpu):
    """A NaN column amount in one layer: the NaN positions of every output field and the finite values elsewhere must be the
    compiled reference's (fixtures nan_column / nan_sgl_column).  The clamped brackets of the fast loops return 0 for a NaN or
    negative amplitude; such lines take the general loop (lines_device.hpp line_records).  Sparse channels
    (one wavenumber per lane) and a 700-point grid (two per lane, far field, physics pass)."""
    for name, kind, tol in (("nan_column", 8, RTOL), ("nan_sgl_column", 4, SGL_VS_SGL)):
        g = Golden(name, workdir)
        for i, (pr, exp) in enumerate(zip(g.profiles, g.expected)):
            rt = api.MonoRTM(g.tape3, pr.wn[0], pr.wn[-1], real_kind=kind)
            compare_nan_aware(rt.run([pr])[0], exp, rtol=tol, what=f"{name}[{i}] real_kind={kind}", rad_floor=1e-30 if kind == 4 else 0.0)
            rt.close()


def test_hip_batch_equals_single(workdir, gpu):
    g = Golden("cloud_updown", workdir)
    rt = api.MonoRTM(g.tape3, g.profiles[0].wn[0], g.profiles[0].wn[-1])
    batch = rt.run(g.profiles)
    for i, (pr, exp) in enumerate(zip(g.profiles, g.expected)):
        compare(batch[i], exp, rtol=RTOL, what=f"batch[{i}]")
        single = rt.run([pr])[0]
        assert np.array_equal(single.o, batch[i].o) and np.array_equal(single.tb, batch[i].tb)
    rt.close()


def test_hip_matches_oracle_random_batch(workdir, gpu):
    """Fresh seeded inputs (not in the golden set): ragged layer counts, cloud, both viewing geometries."""
    from oracle.pyoracle import Oracle

    rec = synth.synthetic_lines(400, seed=4242, sdep_frac=0.2, lc_frac=0.5)
    t3 = f"{workdir}/TAPE3_rand"
    tape3.write_tape3(t3, rec)
    wn = synth.c2_channels(40, seed=9)
    profs = [synth.perturbed_profile(100 + i, wn, nlay=nl, cloud=(i % 2 == 0), irt=(1 if i % 3 == 0 else 3))
             for i, nl in enumerate((64, 40, 17, 64, 33, 5))]
    rt = api.MonoRTM(t3, wn[0], wn[-1])
    orc = Oracle(t3, wn[0], wn[-1])
    # same scalar options for the whole batch, but irt / tmpsfc / emissivity vary per profile
    got = rt.run(profs)
    for i, pr in enumerate(profs):
        compare(got[i], orc.run(pr), rtol=RTOL, what=f"random[{i}] nlay={pr.nlay}")
    rt.close()


def test_device_batch_matches_host_path(workdir, gpu):
    g = Golden("c2_base", workdir)
    rt = api.MonoRTM(g.tape3, g.profiles[0].wn[0], g.profiles[0].wn[-1])
    db = api.DeviceBatch(rt, g.profiles)
    db.step()
    db.check()
    compare(db.dumps(g.profiles)[0], g.expected[0], rtol=RTOL, what="device batch")
    rt.close()


def test_error_paths(workdir, gpu):
    g = Golden("cntnm_factors", workdir)
    pr = g.profiles[0]
    with pytest.raises(api.MonoRTMError) as e:
        api.MonoRTM(workdir + "/does_not_exist", 1.0, 2.0)
    assert e.value.code == 1
    rt = api.MonoRTM(g.tape3, pr.wn[0], pr.wn[-1])
    cold = synth.Profile(wn=pr.wn, p=pr.p, t=np.full_like(pr.t, 50.0), tz=pr.tz, wkl=pr.wkl, wbrodl=pr.wbrodl, clw=pr.clw)
    with pytest.raises(api.MonoRTMError) as e:
        rt.run([cold])
    assert e.value.code == 4  # reference: STOP in TIPS (tips_2003.f90:277)
    with pytest.raises(api.MonoRTMError) as e:
        rt.modm([pr], ixsect=1)
    assert e.value.code == 3
    rt.run([pr])  # context still usable after an error
    rt.close()


def test_edge_empty_line_list_and_wide_grid(workdir, gpu):
    """Header-only TAPE3 (continuum + cloud only) on a 20000-point grid (4 tiles wide enough to need the 256-thread
    line kernel and the large continuum grid), against the oracle."""
    from oracle.pyoracle import Oracle

    t3 = f"{workdir}/TAPE3_hdr_only"
    tape3.write_tape3(f"{workdir}/TAPE3_tmp10", synth.synthetic_lines(10))
    open(t3, "wb").write(open(f"{workdir}/TAPE3_tmp10", "rb").read()[: 1664 + 8])
    a = synth.standard_atmosphere(3)
    wn = 0.5 + 0.0025 * np.arange(20000)
    clw = np.array([0.02, 0.0, 0.0])
    pr = synth.Profile(wn=wn, p=a["p"], t=a["t"], tz=a["tz"], wkl=a["wkl"], wbrodl=a["wbrodl"], clw=clw, irt=3, dvset=0.0025)
    rt = api.MonoRTM(t3, wn[0], wn[-1])
    assert rt.line_count(0) == 0
    got = rt.run([pr])[0]
    assert np.all(got.o_by_mol == 0.0)
    compare(got, Oracle(t3, wn[0], wn[-1]).run(pr), rtol=RTOL, what="empty line list, 20000 wn")
    rt.close()


def test_temperature_stop_does_not_depend_on_the_line_window(workdir, gpu):
    """MODM calls TIPS_2003 for every layer and all NMOL molecules (src/modm.f90:250) and any QT_* routine returns -1 outside
    70-3000 K -> STOP (src/tips_2003.f90:272-277), whatever the line file holds: a header-only TAPE3, a window without lines
    and a single bad layer in the middle of a profile must all give MONORTM_ETEMP; the oracle agrees."""
    from oracle.pyoracle import Oracle, OracleError

    hdr = f"{workdir}/TAPE3_hdr_only_t"
    tape3.write_tape3(f"{workdir}/TAPE3_tmp10t", synth.synthetic_lines(10))
    open(hdr, "wb").write(open(f"{workdir}/TAPE3_tmp10t", "rb").read()[: 1664 + 8])
    far = f"{workdir}/TAPE3_far_lines"
    tape3.write_tape3(far, synth.synthetic_lines(20, seed=3, vlo=40.0, vhi=54.0))   # nothing within 25 cm-1 of the channels
    a = synth.standard_atmosphere(5)
    wn = np.array([1.0, 2.5, 6.0])
    for t3 in (hdr, far):
        for bad_t, lay in ((50.0, 0), (69.999, 2), (3000.5, 4)):
            t = a["t"].copy()
            t[lay] = bad_t
            pr = synth.Profile(wn=wn, p=a["p"], t=t, tz=a["tz"], wkl=a["wkl"], wbrodl=a["wbrodl"], clw=a["clw"], irt=3)
            rt = api.MonoRTM(t3, wn[0], wn[-1])
            with pytest.raises(api.MonoRTMError) as e:
                rt.run([pr])
            assert e.value.code == 4, (t3, bad_t)
            ok = synth.Profile(wn=wn, p=a["p"], t=np.clip(t, 70.0, 3000.0), tz=a["tz"], wkl=a["wkl"], wbrodl=a["wbrodl"], clw=a["clw"], irt=3)
            rt.run([ok])     # exactly 70 K / 3000 K are inside (".lt.70. .OR. .gt.3000.")
            rt.close()
            with pytest.raises(OracleError):
                Oracle(t3, wn[0], wn[-1]).run(pr)


def test_edge_many_lines_few_channels_sliced(workdir, gpu):
    """Single profile, 3 channels, 5000 lines: the line list is sliced over several workgroups per layer; the
    sliced sums must match the oracle like the unsliced ones."""
    from oracle.pyoracle import Oracle

    t3 = f"{workdir}/TAPE3_5000"
    tape3.write_tape3(t3, synth.synthetic_lines(5000, seed=31, sdep_frac=0.1, lc_frac=0.3))
    a = synth.standard_atmosphere(5, ztop_km=40)
    wn = np.array([0.7417, 1.9, 22.0])
    pr = synth.Profile(wn=wn, p=a["p"], t=a["t"], tz=a["tz"], wkl=a["wkl"], wbrodl=a["wbrodl"], clw=a["clw"], irt=1,
                       tmpsfc=280.0, emiss=np.full(3, 0.9), reflc=np.full(3, 0.1))
    rt = api.MonoRTM(t3, wn[0], wn[-1])
    compare(rt.run([pr])[0], Oracle(t3, wn[0], wn[-1]).run(pr), rtol=RTOL, what="sliced line list")
    rt.close()


def test_argument_errors(workdir, gpu):
    g = Golden("cntnm_factors", workdir)
    pr = g.profiles[0]
    rt = api.MonoRTM(g.tape3, pr.wn[0], pr.wn[-1])
    desc = synth.Profile(wn=pr.wn[::-1].copy(), p=pr.p, t=pr.t, tz=pr.tz, wkl=pr.wkl, wbrodl=pr.wbrodl, clw=pr.clw)
    with pytest.raises(api.MonoRTMError) as e:
        rt.run([desc])                      # descending wavenumbers
    assert e.value.code == 6
    few = synth.Profile(wn=pr.wn, p=pr.p, t=pr.t, tz=pr.tz, wkl=pr.wkl[:, :5], wbrodl=pr.wbrodl, clw=pr.clw)
    with pytest.raises(api.MonoRTMError) as e:
        rt.run([few])                       # NMOL < 7: LINES reads WK(1:7) (modm.f90:313)
    assert e.value.code == 6
    rt.close()


# ---- single precision (real_kind = 4: the reference's "sgl" build, BASELINE config 5) --------------------------------
# The sgl reference accumulates in REAL*4; the HIP path keeps double wherever that is free (preparation, continuum,
# recurrences), so it sits between the two reference builds.  Tolerances: 2e-4 against the sgl reference (its own
# float noise, the same bound the REAL*4 Fortran caller test uses), 5e-5 against the dbl reference (the float Lorentz
# loop subtracts the 25 cm-1 pedestal in float: far-wing terms of a molecule with few lines cancel to ~1e-5 relative).
SGL_VS_SGL = 2e-4
SGL_VS_DBL = 5e-5
SGL_LONG_SUMS = {("sgl_real_like", 2): 2e-3}   # 1200-point grid x 3300 lines: (fixture, profile) -> tolerance against the sgl reference


def test_real4_matches_sgl_reference(workdir, gpu):
    for name in golden_names(single_precision=True):
        g = Golden(name, workdir)
        pr0 = g.profiles[0]
        rt = api.MonoRTM(g.tape3, pr0.wn[0], pr0.wn[-1], real_kind=4)
        for i, (pr, exp) in enumerate(zip(g.profiles, g.expected)):  # one call per profile: the scalar options differ
            got = rt.run([pr])[0]
            assert got.o.dtype == np.float32 and got.tb.dtype == np.float32
            if (name, i) in SGL_LONG_SUMS:
                # the sgl reference adds thousands of REAL*4 terms of both signs per (wavenumber, layer) here - its own rounding
                # noise reaches 1e-3 of the sum.  Held loosely to it, and tightly to the double-precision restatement reading the
                # file as the sgl build does (the HIP path keeps its sums in double where that is free)
                from oracle.pyoracle import Oracle

                compare(got, exp, rtol=SGL_LONG_SUMS[(name, i)], what=f"real4 {name}[{i}] vs the sgl reference")
                orc = Oracle(g.tape3, pr0.wn[0], pr0.wn[-1], real_kind=4)
                # (4 x SGL_VS_DBL: the mis-walked O2 records of this file add terms of both signs that cancel to a few per cent
                # in a cell - float terms cannot do better; observed 1.0e-4 where the sgl reference itself is 7.8e-4 off)
                compare(got, orc.run(pr), rtol=4 * SGL_VS_DBL, what=f"real4 {name}[{i}] vs the oracle (sgl file rules, double arithmetic)", rad_floor=1e-30)
                orc.close()
                continue
            compare(got, exp, rtol=SGL_VS_SGL, what=f"real4 {name}[{i}]")
        rt.close()


# negative_strength: per-molecule optical depths of both signs cancel in the layer total O to a few per cent of its terms - a
# REAL*4 total cannot be 5e-5 of the REAL*8 one there (observed 1.9e-4 on O, every other field <= 9e-7).  Its single-precision
# twin sgl_negative_strength holds the real_kind = 4 kernels to the sgl reference instead (test_real4_matches_sgl_reference).
# real_like: its line file holds a coupling record as the first record of a block, which the two builds of the reference FILE
# differently (molecule 4 in "dbl", 24 in "sgl": line_table.cpp mol0_owner) - a real_kind = 4 context follows the sgl rule and
# differs from the dbl fixture in N2O by 12 %.  sgl_real_like is its twin.
REAL4_VS_DBL_SKIP = {"negative_strength", "real_like"}


@pytest.mark.parametrize("name", [n for n in golden_names() if n not in REAL4_VS_DBL_SKIP])
def test_real4_close_to_dbl_reference(name, workdir, gpu):
    g = Golden(name, workdir)
    pr0 = g.profiles[0]
    rt = api.MonoRTM(g.tape3, pr0.wn[0], pr0.wn[-1], real_kind=4)
    for i, (pr, exp) in enumerate(zip(g.profiles, g.expected)):  # one call per profile: the scalar options differ
        compare(rt.run([pr])[0], exp, rtol=SGL_VS_DBL, what=f"real4 {name}[{i}]", rad_floor=1e-30)
    rt.close()


def test_real4_device_batch(workdir, gpu):
    g = Golden("cloud_updown", workdir)
    rt = api.MonoRTM(g.tape3, g.profiles[0].wn[0], g.profiles[0].wn[-1], real_kind=4)
    db = api.DeviceBatch(rt, g.profiles)
    db.step()
    db.check()
    host = rt.run(g.profiles)
    for i, d in enumerate(db.dumps(g.profiles)):
        assert np.array_equal(d.o, host[i].o) and np.array_equal(d.rad, host[i].rad)
        compare(d, g.expected[i], rtol=SGL_VS_DBL, what=f"real4 device batch[{i}]")
    rt.close()


def test_graph_replay_equals_stream_launches(workdir, gpu):
    """The step recorded into a HIP graph produces bit-identical results to the three stream launches."""
    import torch

    g = Golden("cloud_updown", workdir)
    rt = api.MonoRTM(g.tape3, g.profiles[0].wn[0], g.profiles[0].wn[-1])
    db = api.DeviceBatch(rt, g.profiles)
    db.step()
    torch.cuda.synchronize()
    ref = db.dumps(g.profiles)
    db.capture()
    for t in (db.O, db.OBM, db.OC, db.RAD, db.TB, db.TMR):
        t.zero_()
    db.replay()
    torch.cuda.synchronize()
    db.check()
    for i, d in enumerate(db.dumps(g.profiles)):
        for k in ("o", "o_by_mol", "oc", "o_clw", "rup", "rdn", "trtot", "rad", "tb", "tmr"):
            assert np.array_equal(getattr(d, k), getattr(ref[i], k)), k
        compare(d, g.expected[i], rtol=RTOL, what=f"graph replay[{i}]")
    rt.close()


def test_graph_replay_dense_grid(workdir, gpu):
    """A dense grid recorded into a HIP graph: the step then holds physics_kernel, far_plan_kernel on the context's side stream
    (forked from and joined to the captured stream with two events), far_kernel per level, lines_kernel, the slice reduction,
    finish and rtm kernels.  Replays are bit-identical to the stream launches."""
    import torch

    t3 = f"{workdir}/TAPE3_graphdense"
    tape3.write_tape3(t3, synth.synthetic_lines(2500, seed=909, vlo=0.05, vhi=54.9))
    wn = 4.0 + 0.004 * np.arange(2300)
    a = synth.standard_atmosphere(3, ztop_km=25)
    profs = [synth.Profile(wn=wn, p=a["p"], t=a["t"] + dt, tz=a["tz"] + dt, wkl=a["wkl"], wbrodl=a["wbrodl"], clw=a["clw"], irt=3, dvset=0.004)
             for dt in (0.0, 3.0)]
    rt = api.MonoRTM(t3, wn[0], wn[-1])
    db = api.DeviceBatch(rt, profs)
    db.step()
    torch.cuda.synchronize()
    ref = db.dumps(profs)
    db.capture()
    for _ in range(2):
        for t in (db.O, db.OBM, db.OC, db.RAD, db.TB, db.TMR):
            t.zero_()
        db.replay()
        torch.cuda.synchronize()
        db.check()
        for i, d in enumerate(db.dumps(profs)):
            for k in ("o", "o_by_mol", "rad", "tb"):
                assert np.array_equal(getattr(d, k), getattr(ref[i], k)), k
    rt.close()


@pytest.mark.parametrize("nwn,nlay", [(1, 1), (2, 3), (63, 2), (64, 24), (65, 5), (127, 2), (128, 25), (129, 3), (255, 2), (256, 4),
                                      (257, 2), (513, 3), (50, 200), (600, 5), (1100, 7), (1600, 11)])
def test_shape_sweep_against_oracle(nwn, nlay, workdir, gpu):
    """(The last three: several tiles per layer with a number of (layer, slice) groups that is no multiple of 8 - the XCD-aware
    block placement of lines_kernel takes its remainder path.)
    Every kernel configuration boundary (1 / 2 wavenumbers per lane, 1 / 2 / 4 waves, partial last tile, layer groups of
    the radiance kernel, a 200-layer profile) with coupled, speed-dependent and plain lines, both kinds of context."""
    from oracle.pyoracle import Oracle

    t3 = f"{workdir}/TAPE3_sweep"
    tape3.write_tape3(t3, synth.synthetic_lines(300, seed=77, sdep_frac=0.15, lc_frac=0.4))
    rng = np.random.default_rng(nwn * 1000 + nlay)
    wn = np.sort(rng.uniform(0.2, 40.0, nwn))
    a = synth.standard_atmosphere(nlay, ztop_km=60)  # up to the mesosphere: Voigt shapes at the top
    clw = np.zeros(nlay)
    clw[0] = 0.03
    up = nlay % 2 == 1
    pr = synth.Profile(wn=wn, p=a["p"], t=a["t"], tz=a["tz"], wkl=a["wkl"], wbrodl=a["wbrodl"], clw=clw, irt=1 if up else 3,
                       **(dict(tmpsfc=285.0, emiss=np.full(nwn, 0.8), reflc=np.full(nwn, 0.2)) if up else {}))
    exp = Oracle(t3, wn[0], wn[-1]).run(pr)
    rt = api.MonoRTM(t3, wn[0], wn[-1])
    compare(rt.run([pr])[0], exp, rtol=RTOL, what=f"sweep nwn={nwn} nlay={nlay}")
    rt.close()
    rt4 = api.MonoRTM(t3, wn[0], wn[-1], real_kind=4)
    compare(rt4.run([pr])[0], exp, rtol=SGL_VS_DBL, what=f"sweep real4 nwn={nwn} nlay={nlay}", rad_floor=1e-30)
    rt4.close()


@pytest.mark.parametrize("nwn,nprof", [(100, 1), (200, 1), (200, 160)])
def test_sounder_channels_real4_full_class(nwn, nprof, workdir, gpu):
    """Single precision, channels below 6.5 cm-1 (configs[4]'s sounder shape): the two-resonance lines below 18.5 cm-1 are within
    reach of EVERY channel and take the FULL class of the float loops (t = d d+ + HW^2 form, lines_device.hpp) - in the
    two-wavenumber tile (100 / 200 channels of a few profiles) and in the four-wavenumber tile of large batches (160 profiles x
    64 layers >= 8192 states).  Held to the dbl oracle like every real_kind = 4 result."""
    from oracle.pyoracle import Oracle

    t3 = f"{workdir}/TAPE3_full"
    tape3.write_tape3(t3, synth.synthetic_lines(400, seed=11))
    wn = synth.c2_channels(nwn, lo=0.3, hi=6.5)
    profs = []
    for i in range(nprof):
        a = synth.standard_atmosphere(64 if nprof > 1 else 20, ztop_km=30)
        clw = np.zeros(len(a["p"]))
        clw[1] = 0.02
        profs.append(synth.Profile(wn=wn, p=a["p"] * (1.0 + 0.001 * i), t=a["t"] + 0.01 * i, tz=a["tz"] + 0.01 * i, wkl=a["wkl"],
                                   wbrodl=a["wbrodl"], clw=clw, irt=3))
    rt4 = api.MonoRTM(t3, wn[0], wn[-1], real_kind=4)
    got = rt4.run(profs)
    rt4.close()
    orc = Oracle(t3, wn[0], wn[-1])
    for i in sorted({0, nprof // 2, nprof - 1}):
        compare(got[i], orc.run(profs[i]), rtol=SGL_VS_DBL, what=f"sounder real4 nwn={nwn} profile {i} of {nprof}", rad_floor=1e-30)


def test_implausibly_strong_lines_stay_exact(workdir, gpu):
    """The fast loops of generic molecules test the 25 cm-1 rule with the [0, 1] clamp of an FMA (lines_asm.hpp); a line whose
    peak a2 / HW^2 could come near 1 (ten orders of magnitude above any physical line strength) is routed to the general loop
    instead and must give the oracle's numbers."""
    from oracle.pyoracle import Oracle

    rec = synth.synthetic_lines(60, seed=21)
    rec.sp = np.asarray(rec.sp) * np.where(np.arange(len(rec.sp)) % 3 == 0, 1e17, 1.0)   # every third line: S x 1e17
    t3 = f"{workdir}/TAPE3_strong"
    tape3.write_tape3(t3, rec)
    strong = np.asarray(rec.vnu)[np.arange(len(rec.sp)) % 3 == 0]
    wn = np.unique(np.concatenate([strong[strong < 40.0], synth.c2_channels(20)]))   # channels ON the strong lines' centres
    a = synth.standard_atmosphere(12, ztop_km=40)
    pr = synth.Profile(wn=wn, p=a["p"], t=a["t"], tz=a["tz"], wkl=a["wkl"] * 1e-17, wbrodl=a["wbrodl"], clw=np.zeros(12), irt=3)
    rt = api.MonoRTM(t3, wn[0], wn[-1])
    compare(rt.run([pr])[0], Oracle(t3, wn[0], wn[-1]).run(pr), rtol=RTOL, what="strong lines")
    rt.close()


def test_c_example_calls_the_abi(workdir, gpu):
    """examples/call_abi.c: the C ABI from plain C (gcc, no Python / Fortran in the caller) gives the numbers of the
    oracle for the same profile."""
    import os
    import re
    import subprocess

    from common import ROOT
    from monortm_amd import _build
    from oracle.pyoracle import Oracle

    exe = f"{workdir}/call_abi"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "call_abi.c"),
                           "-L", _build.LIBDIR, "-lmonortm_hip", f"-Wl,-rpath,{_build.LIBDIR}", "-lm", "-o", exe])
    t3 = f"{workdir}/TAPE3_cex"
    tape3.write_tape3(t3, synth.synthetic_lines(200, seed=5))
    out = subprocess.run([exe, t3], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    rows = [[float(x) for x in re.findall(r"[-+]?\d+\.\d+(?:e[-+]\d+)?", ln)] for ln in out.stdout.splitlines() if "TB" in ln]
    assert len(rows) == 6
    # the same profile, rebuilt here
    wn = np.array([0.7417, 0.7939, 1.0474, 1.7, 3.0, 5.0])
    tz = 288.0 - 13.0 * np.arange(5)
    z = 2.0 * np.arange(4) + 1.0
    dp = 1013.0 * (np.exp(-(z - 1.0) / 7.5) - np.exp(-(z + 1.0) / 7.5))
    air = 2.1e25 * dp / 1013.0
    vmr = np.array([0.0, 4.0e-4, 3.0e-7, 3.2e-7, 1.5e-7, 1.7e-6, 0.209])
    wkl = vmr[None, :] * air[:, None]
    wkl[:, 0] = 0.01 * np.exp(-z / 2.0) * air
    pr = synth.Profile(wn=wn, p=1013.0 * np.exp(-z / 7.5), t=0.5 * (tz[:-1] + tz[1:]), tz=tz, wkl=wkl, wbrodl=0.781 * air,
                       clw=np.zeros(4), irt=3, tmpsfc=288.0, emiss=np.ones(6), reflc=np.zeros(6))
    exp = Oracle(t3, wn[0], wn[-1]).run(pr)
    got = np.array(rows)
    assert np.allclose(got[:, 1], exp.tb, rtol=0, atol=6e-6) and np.allclose(got[:, 2], exp.tmr, rtol=0, atol=6e-6)
    assert np.allclose(got[:, 3], exp.o.sum(axis=0), rtol=1e-8)


@pytest.mark.parametrize("v1,dv,nwn,nlines", [(10.0, 0.005, 700, 4000), (0.4, 0.002, 1200, 4000), (38.0, 0.01, 513, 4000),
                                              (12.0, 0.005, 600, 400), (20.0, 0.004, 520, 150), (3.0, 0.003, 900, 700)])
def test_far_field_dense_grid(v1, dv, nwn, nlines, workdir, gpu):
    """Dense grids take the far-field path of the line kernel (lines at >= 4 tile half-widths from the tile centre are
    summed through 26 moments per molecule instead of one Lorentzian each).  The truncation is below 1e-14 of a term, so the
    result is held to 1e-10 of the oracle here - four orders tighter than the product tolerance - for generic molecules,
    uncoupled O2, CO2 (quadratic pedestal) and two-resonance lines near 0 cm-1."""
    from oracle.pyoracle import Oracle

    t3 = f"{workdir}/TAPE3_far"
    # few lines: several molecule runs share a chunk of 256 lines (moment slots by molecule parity, waves that straddle runs)
    rec = synth.synthetic_lines(nlines, seed=int(v1 * 10), vlo=0.05, vhi=54.9)
    ibrd = int(nlines == 700)  # one case with species-by-species broadening data (the IBRD instantiation of the kernel)
    if ibrd:
        rng = np.random.default_rng(11)
        n = len(rec.vnu)
        rec.brd_flg = (rng.random((n, 7)) < 0.3).astype(np.int32)
        dat = np.zeros((n, 21), np.float32)
        dat[:, 0::3], dat[:, 1::3], dat[:, 2::3] = rng.uniform(0.03, 0.15, (n, 7)), rng.uniform(0.4, 0.8, (n, 7)), rng.uniform(-0.004, 0.004, (n, 7))
        rec.brd_dat = dat
    tape3.write_tape3(t3, rec)
    wn = v1 + dv * np.arange(nwn)
    a = synth.standard_atmosphere(3, ztop_km=12)
    pr = synth.Profile(wn=wn, p=a["p"], t=a["t"], tz=a["tz"], wkl=a["wkl"], wbrodl=a["wbrodl"], clw=a["clw"], irt=3, dvset=dv,
                       ibrd=ibrd)
    exp = Oracle(t3, wn[0], wn[-1]).run(pr)
    rt = api.MonoRTM(t3, wn[0], wn[-1])
    errs = compare(rt.run([pr])[0], exp, rtol=1e-10, what=f"far field v1={v1} dv={dv} nwn={nwn} nlines={nlines}")
    rt.close()
    assert errs["o_by_mol"] < 1e-10


@pytest.mark.parametrize("v1,dv,nwn,nlines", [(10.0, 0.005, 700, 4000), (0.4, 0.002, 1200, 400)])
def test_far_field_dense_grid_real4(v1, dv, nwn, nlines, workdir, gpu):
    """The single-precision build uses the same far-field moments (formed in double, added to the float sums)."""
    from oracle.pyoracle import Oracle

    t3 = f"{workdir}/TAPE3_far4"
    tape3.write_tape3(t3, synth.synthetic_lines(nlines, seed=int(v1 * 10) + 1, vlo=0.05, vhi=54.9))
    wn = v1 + dv * np.arange(nwn)
    a = synth.standard_atmosphere(3, ztop_km=12)
    pr = synth.Profile(wn=wn, p=a["p"], t=a["t"], tz=a["tz"], wkl=a["wkl"], wbrodl=a["wbrodl"], clw=a["clw"], irt=3, dvset=dv)
    exp = Oracle(t3, wn[0], wn[-1]).run(pr)
    rt = api.MonoRTM(t3, wn[0], wn[-1], real_kind=4)
    compare(rt.run([pr])[0], exp, rtol=SGL_VS_DBL, what=f"far field real4 v1={v1} dv={dv} nwn={nwn}", rad_floor=1e-30)
    rt.close()


@pytest.mark.parametrize("tile_waves", ["auto", 1, 2])
@pytest.mark.parametrize("v1,dv,nwn,nlines,levels", [(8.0, 0.004, 2100, 3000, None), (0.4, 0.002, 2600, 400, None), (30.0, 0.005, 1537, 150, None),
                                                     (3.0, 0.003, 4000, 40, None), (12.0, 0.005, 2100, 3000, 1), (12.0, 0.005, 2100, 3000, 2),
                                                     (2.0, 0.001, 2049, 3000, 4)])
def test_far_kernel_dense_grid(v1, dv, nwn, nlines, levels, tile_waves, workdir, gpu):
    """Grids of >= 4 tiles of 512 wavenumbers: the far lines of every tile come through far_kernel (far_kernel.hip: expanded once by the
    widest interval - tile, pair of tiles, four, eight - for which they are far; a child adds its parent's series re-expanded about
    its own centre) and lines_kernel walks only the runs far_plan_kernel leaves.  Held to 1e-10 of the oracle like the in-kernel far
    field, and to 1e-11 of the result with far_levels = 0 (the far field formed inside lines_kernel): few lines per molecule (a
    molecule whose only lines in a tile's window are far ones is written from the series alone), a last tile of ONE wavenumber
    (1537, 2049: no far field for it nor for the intervals that hold it), every level count, CO2 / O2 / two-resonance lines."""
    from oracle.pyoracle import Oracle

    t3 = f"{workdir}/TAPE3_fark"
    tape3.write_tape3(t3, synth.synthetic_lines(nlines, seed=int(v1 * 10) + nlines, vlo=0.05, vhi=54.9))
    wn = v1 + dv * np.arange(nwn)
    a = synth.standard_atmosphere(3, ztop_km=30)
    pr = synth.Profile(wn=wn, p=a["p"], t=a["t"], tz=a["tz"], wkl=a["wkl"], wbrodl=a["wbrodl"], clw=a["clw"], irt=3, dvset=dv)
    exp = Oracle(t3, wn[0], wn[-1]).run(pr)
    rt = api.MonoRTM(t3, wn[0], wn[-1])
    # (tiles of 512 wavenumbers unless the line list is dense enough for smaller ones: 1 / 2 force the one- and two-wave tiles of 128
    # and 256 - the one-wave tile is an instantiation of its own that has no far field but far_kernel's)
    rt.set_option("tile_waves", tile_waves)
    if levels is not None:
        rt.set_option("far_levels", levels)
    got = rt.run([pr, pr])[1]
    errs = compare(got, exp, rtol=1e-10, what=f"far kernel v1={v1} dv={dv} nwn={nwn} nlines={nlines} levels={levels} tile_waves={tile_waves}")
    assert errs["o_by_mol"] < 1e-10
    rt.set_option("far_levels", 0)
    rt.set_option("tile_waves", "auto")
    ref = rt.run([pr])[0]
    rt.set_option("tile_waves", tile_waves)
    rt.set_option("nslice", 3)   # (the series joins in the slice that holds a molecule's last candidate; far-only molecules in slice 0)
    rt.set_option("far_levels", "auto" if levels is None else levels)
    sl = rt.run([pr])[0]
    rt.close()
    scale = np.abs(ref.o_by_mol).max(axis=2, keepdims=True) + 1e-300
    assert np.max(np.abs(got.o_by_mol - ref.o_by_mol) / scale) < 1e-11
    assert np.max(np.abs(sl.o_by_mol - ref.o_by_mol) / scale) < 1e-11
    np.testing.assert_allclose(got.tb, ref.tb, rtol=1e-10)


@pytest.mark.parametrize("v1,dv,nwn,top_km", [(2000.0, 0.002, 2100, 30.0), (900.0, 0.0004, 2600, 70.0)])
def test_far_kernel_infrared_grid(v1, dv, nwn, top_km, workdir, gpu):
    """Dense infrared grids: Doppler widths are ~1e-3 cm-1 there, so the plan's guard (no far line within 100 Doppler widths of any
    wavenumber, modm.f90:427) matters - the second case has tiles of 0.2 cm-1 for which it fails at the tile level and thin upper
    layers with Voigt candidates: lines_kernel then keeps those lines (and its own far field); both paths against the oracle."""
    from oracle.pyoracle import Oracle

    t3 = f"{workdir}/TAPE3_farir"
    tape3.write_tape3(t3, synth.synthetic_lines(3000, seed=int(v1), vlo=v1 - 26.0, vhi=v1 + dv * nwn + 26.0))
    wn = v1 + dv * np.arange(nwn)
    a = synth.standard_atmosphere(4, ztop_km=top_km)
    pr = synth.Profile(wn=wn, p=a["p"], t=a["t"], tz=a["tz"], wkl=a["wkl"], wbrodl=a["wbrodl"], clw=a["clw"], irt=3, dvset=dv)
    exp = Oracle(t3, wn[0], wn[-1]).run(pr)
    rt = api.MonoRTM(t3, wn[0], wn[-1])
    compare(rt.run([pr])[0], exp, rtol=1e-9, what=f"far kernel infrared v1={v1} dv={dv}")
    rt.set_option("far_levels", 0)
    compare(rt.run([pr])[0], exp, rtol=1e-9, what=f"in-kernel far field infrared v1={v1} dv={dv}")
    rt.close()


@pytest.mark.parametrize("tile_waves", ["auto", 1])
def test_far_kernel_unphysical_inputs(tile_waves, workdir, gpu):
    """Dense grid with a third of the line strengths NEGATIVE (the far field is linear in the amplitude: a negative line subtracts, as
    src/modm.f90:432 does) and, second profile, a NaN column amount in one layer (that layer's state is not finite: no far lines,
    every line kept, NaN where the reference puts NaN): values and NaN positions against the oracle, which is pinned to the compiled
    reference on the fixtures negative_strength / nan_column."""
    from oracle.pyoracle import Oracle

    rec = synth.synthetic_lines(1800, seed=2718, vlo=0.05, vhi=54.9)
    rec.sp = np.where(np.arange(len(rec.sp)) % 3 == 1, -rec.sp, rec.sp)
    t3 = f"{workdir}/TAPE3_farneg"
    tape3.write_tape3(t3, rec)
    wn = 7.0 + 0.004 * np.arange(2100)
    a = synth.standard_atmosphere(3, ztop_km=20)
    wkl_nan = a["wkl"].copy()
    wkl_nan[1, 2] = np.nan
    profs = [synth.Profile(wn=wn, p=a["p"], t=a["t"], tz=a["tz"], wkl=a["wkl"], wbrodl=a["wbrodl"], clw=a["clw"], irt=3, dvset=0.004),
             synth.Profile(wn=wn, p=a["p"], t=a["t"], tz=a["tz"], wkl=wkl_nan, wbrodl=a["wbrodl"], clw=a["clw"], irt=3, dvset=0.004)]
    orc = Oracle(t3, wn[0], wn[-1])
    rt = api.MonoRTM(t3, wn[0], wn[-1])
    rt.set_option("tile_waves", tile_waves)
    out = rt.run(profs)
    rt.close()
    compare(out[0], orc.run(profs[0]), rtol=1e-8, what=f"far kernel, negative strengths, tile_waves={tile_waves}")
    compare_nan_aware(out[1], orc.run(profs[1]), rtol=1e-8, what=f"far kernel, NaN column, tile_waves={tile_waves}")


@pytest.mark.parametrize("tile_waves", ["auto", 1, 2])
def test_far_kernel_real4_and_batch(tile_waves, workdir, gpu):
    """far_kernel in the single-precision build (amplitudes carry the column amount, sums formed in double) and for a batch whose
    profiles have different numbers of layers; with every tile size (1: the one-wave tile that has far_kernel's far field alone)."""
    from oracle.pyoracle import Oracle

    t3 = f"{workdir}/TAPE3_fark4"
    tape3.write_tape3(t3, synth.synthetic_lines(2000, seed=4242, vlo=0.05, vhi=54.9))
    wn = 5.0 + 0.004 * np.arange(2300)
    profs = []
    for nl in (2, 4, 3):
        a = synth.standard_atmosphere(nl, ztop_km=25)
        profs.append(synth.Profile(wn=wn, p=a["p"], t=a["t"], tz=a["tz"], wkl=a["wkl"], wbrodl=a["wbrodl"], clw=a["clw"], irt=3, dvset=0.004))
    orc = Oracle(t3, wn[0], wn[-1])
    for rk, tol in ((8, 1e-10), (4, SGL_VS_DBL)):
        rt = api.MonoRTM(t3, wn[0], wn[-1], real_kind=rk)
        rt.set_option("tile_waves", tile_waves)
        out = rt.run(profs)
        rt.close()
        for p, g in zip(profs, out):
            compare(g, orc.run(p), rtol=tol, what=f"far kernel batch real_kind={rk} nlay={p.nlay} tile_waves={tile_waves}", rad_floor=1e-30 if rk == 4 else 0.0)


_PHYS_CHILD = r"""
import sys, numpy as np
from monortm_amd import api, synth, tape3
t3, out, ibrd, rk = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
wn = 2.0 + 0.004 * np.arange(2100)
a = synth.standard_atmosphere(4, ztop_km=40)
pr = synth.Profile(wn=wn, p=a["p"], t=a["t"], tz=a["tz"], wkl=a["wkl"], wbrodl=a["wbrodl"], clw=a["clw"], irt=3, dvset=0.004, ibrd=ibrd)
rt = api.MonoRTM(t3, wn[0], wn[-1], real_kind=rk)
d = rt.run([pr, pr])[1]
np.savez(out, o=d.o, obm=d.o_by_mol, tb=d.tb, rad=d.rad)
"""


@pytest.mark.parametrize("ibrd,real_kind", [(0, 8), (1, 8), (0, 4)])
def test_physics_pass_equals_in_place(ibrd, real_kind, workdir, gpu):
    """Grids of >= 4 tiles: physics_kernel forms the tile-independent part of every line once per (profile, layer) and
    lines_kernel reads it back.  Bitwise the same as forming it inside every tile (MONORTM_NO_PHYSICS_PASS=1 is read at first
    use, hence child processes), for coupled lines, species broadening and the single-precision build; and within 1e-10
    of the oracle."""
    import os
    import subprocess
    import sys

    from oracle.pyoracle import Oracle

    t3 = f"{workdir}/TAPE3_phys{ibrd}{real_kind}"
    rec = synth.synthetic_lines(3000, seed=77, vlo=0.05, vhi=40.0, lc_frac=0.5)
    if ibrd:
        rng = np.random.default_rng(5)
        n = len(rec.vnu)
        rec.brd_flg = (rng.random((n, 7)) < 0.3).astype(np.int32)
        dat = np.zeros((n, 21), np.float32)
        dat[:, 0::3], dat[:, 1::3], dat[:, 2::3] = rng.uniform(0.03, 0.15, (n, 7)), rng.uniform(0.4, 0.8, (n, 7)), rng.uniform(-0.004, 0.004, (n, 7))
        rec.brd_dat = dat
    tape3.write_tape3(t3, rec)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    # (far_levels = 0: the far field formed inside lines_kernel on both sides - far_kernel, the default with the physics pass,
    # adds the same terms in another order: third child, compared to rounding)
    # (fourth child: far_kernel with the one-wave tiles of 128 wavenumbers forced - their species-broadening and float instantiations)
    for off, far, tw in ((False, "0", None), (True, "0", None), (False, None, None), (False, None, "1")):
        env = dict(os.environ, PYTHONPATH=root)
        env.pop("MONORTM_NO_PHYSICS_PASS", None)
        env.pop("MONORTM_FAR_LEVELS", None)
        env.pop("MONORTM_TILE_WAVES", None)
        if off:
            env["MONORTM_NO_PHYSICS_PASS"] = "1"
        if far is not None:
            env["MONORTM_FAR_LEVELS"] = far
        if tw is not None:
            env["MONORTM_TILE_WAVES"] = tw
        out = f"{workdir}/phys_{ibrd}{real_kind}_{int(off)}{far}{tw}.npz"
        subprocess.run([sys.executable, "-c", _PHYS_CHILD, t3, out, str(ibrd), str(real_kind)], check=True, env=env, timeout=600)
        outs.append(np.load(out))
    for k in ("o", "obm", "tb", "rad"):
        assert np.array_equal(outs[0][k], outs[1][k]), f"{k}: physics pass and in-place path differ"
        for q in (2, 3):
            np.testing.assert_allclose(outs[q][k], outs[0][k], rtol=1e-11 if real_kind == 8 else 2e-6, atol=0,
                                       err_msg=f"{k}: far_kernel (child {q}) and the far field of lines_kernel differ")
    if real_kind == 8:
        wn = 2.0 + 0.004 * np.arange(2100)
        a = synth.standard_atmosphere(4, ztop_km=40)
        pr = synth.Profile(wn=wn, p=a["p"], t=a["t"], tz=a["tz"], wkl=a["wkl"], wbrodl=a["wbrodl"], clw=a["clw"], irt=3, dvset=0.004,
                           ibrd=ibrd)
        exp = Oracle(t3, wn[0], wn[-1]).run(pr)
        assert np.allclose(outs[0]["o"], exp.o, rtol=1e-10, atol=0) and np.allclose(outs[0]["tb"], exp.tb, rtol=1e-10, atol=0)


def test_cross_sections_batch_and_beyond_the_reference_domain(workdir, gpu):
    """IXSECT = 1 on a ragged batch against the oracle (itself pinned to the compiled reference by the xsec_* fixtures), including
    layers whose pressure is BELOW that of the measurements: there the reference overruns convolve()'s work array
    (src/monortm_sub.F90:1758,:1773-1786) while its formulas - evaluated here and in the oracle without the array - select the
    linearly interpolated values.  Double and single precision contexts."""
    import tempfile

    from monortm_amd import xsec
    from oracle.pyoracle import Oracle

    g = Golden("xsec_ccl4_f11_f12", workdir)
    base = g.profiles[0]
    rng = np.random.default_rng(12)
    profs = []
    for i, nl in enumerate((10, 6, 10, 3)):
        scale = (1.0, 0.9, 0.02, 1.0)[i]     # third profile: every layer far below the measurement pressures
        profs.append(synth.Profile(wn=base.wn, p=base.p[:nl] * scale, t=base.t[:nl] + rng.normal(0, 4, nl), tz=base.tz[:nl + 1],
                                   wkl=base.wkl[:nl] * scale, wbrodl=base.wbrodl[:nl] * scale, clw=base.clw[:nl], irt=(1 if i % 2 else 3),
                                   tmpsfc=288.0 if i % 2 else 2.75, emiss=np.full(base.nwn, 0.97 if i % 2 else 1.0),
                                   reflc=np.full(base.nwn, 0.03 if i % 2 else 0.0), xs_names=base.xs_names,
                                   xamnt=base.xamnt[:nl] * scale * rng.uniform(0.5, 2.0, (nl, 3)), xs_dir=g.xs_dir))
    orc = Oracle(g.tape3, base.wn[0], base.wn[-1])
    want = [orc.run(p) for p in profs]
    for rk, tol in ((8, RTOL), (4, 2e-4)):
        rt = api.MonoRTM(g.tape3, base.wn[0], base.wn[-1], real_kind=rk)
        got = rt.run(profs)
        for i in range(len(profs)):
            assert got[i].odxsec is not None and got[i].odxsec.max() > 0
            compare(got[i], want[i], rtol=tol, what=f"xsec batch[{i}] real_kind={rk}", rad_floor=1e-30)
        single = rt.run([profs[1]])[0]
        assert np.array_equal(single.odxsec, got[1].odxsec) and np.array_equal(single.o, got[1].o)
        rt.close()
    # a molecule that FSCDXS does not know / a name that is no cross-section molecule: the reference STOPs in XSREAD
    with pytest.raises(KeyError):
        xsec.load_tables(g.xs_dir, ["NOTAGAS"], 700.0, 900.0)
    with pytest.raises(ValueError):
        xsec.load_tables(g.xs_dir, ["HNO4"], 700.0, 900.0)


@pytest.mark.parametrize("real_kind", [8, 4])
def test_wave_priorities_do_not_touch_the_arithmetic(real_kind, workdir, gpu, monkeypatch):
    """lines_kernel orders the waves of a SIMD by their progress (s_setprio, grids of a few rounds: MONORTM_FAIR overrides the
    choice): scheduling only - every output must be bitwise the same with and without, for one-wave tiles (50 channels) and
    for the two-wavenumber tiles (200 channels)."""
    rec = synth.synthetic_lines(300, seed=31, sdep_frac=0.2, lc_frac=0.5)
    t3 = f"{workdir}/TAPE3_prio_{real_kind}"
    tape3.write_tape3(t3, rec)
    for nwn in (50, 200):
        wn = synth.c2_channels(nwn, seed=9)
        profs = [synth.perturbed_profile(700 + i, wn, nlay=nl, cloud=(i % 2 == 0)) for i, nl in enumerate((64, 33, 64, 7))]
        rt = api.MonoRTM(t3, wn[0], wn[-1], real_kind=real_kind)
        out = {}
        for fair in ("0", "1"):
            rt.set_option("fair", fair)
            out[fair] = rt.run(profs)
        for a, b in zip(out["0"], out["1"]):
            for f in ("o", "o_by_mol", "oc", "o_clw", "rad", "tb", "tmr", "trtot"):
                assert np.array_equal(getattr(a, f), getattr(b, f)), f"{f} differs with wave priorities on (nwn = {nwn})"
        rt.close()
