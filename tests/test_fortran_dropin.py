"""Source-level drop-in test of the Fortran boundary.

examples/harness.f90 calls MODM / CALCTMR / RTM exactly as PROGRAM MONORTM does (reference
src/monortm.f90:557-574).  The SAME source was linked against the reference's own modules to produce
the golden fixtures; here it is linked against monortm_amd/fortran (ISO_C_BINDING shim -> C ABI -> HIP)
and must reproduce them."""
import os
import subprocess

import pytest

from common import RTOL, Golden, compare, golden_names
import numpy as np

from monortm_amd import _build, caseio, synth, tape3

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def harness():
    info = _build.build_fortran_shim()
    assert os.path.exists(info["harness"])
    return info["harness"]


@pytest.mark.parametrize("name", golden_names())
def test_fortran_shim_reproduces_reference(name, workdir, harness):
    g = Golden(name, workdir)
    case = os.path.join(workdir, f"case_{name}.bin")
    out = os.path.join(workdir, f"out_{name}.bin")
    caseio.write_case(case, g.profiles)
    # IXSECT = 1 fixtures: FSCDXS and the xs files are opened by name in the working directory (src/monortm_sub.F90:1341,:1662)
    cwd = getattr(g, "xs_dir", None) or workdir
    r = subprocess.run([harness, case, g.tape3, out], cwd=cwd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "HARNESS_SECONDS" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    dumps = caseio.read_dump(out)
    assert len(dumps) == len(g.expected)
    for i, (got, exp) in enumerate(zip(dumps, g.expected)):
        compare(got, exp, rtol=RTOL, what=f"fortran {name}[{i}]")


def test_single_precision_caller(workdir):
    """A REAL*4 ("sgl" flag set) caller of the drop-in modules: inputs and outputs are default REAL = 4 bytes, the
    GPU computes in f64.  Compared with the reference's own single-precision build; the tolerance is that build's
    rounding noise (it differs from its double-precision sibling by ~1e-5 on these cases)."""
    exe = _build.build_fortran_shim()["harness_sgl"]
    for name in golden_names(single_precision=True):
        g = Golden(name, workdir)
        case = os.path.join(workdir, f"case_{name}.bin")
        out = os.path.join(workdir, f"out_{name}.bin")
        caseio.write_case(case, g.profiles)
        r = subprocess.run([exe, case, g.tape3, out], cwd=workdir, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "HARNESS_SECONDS" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
        for i, (got, exp) in enumerate(zip(caseio.read_dump(out), g.expected)):
            # (sgl_real_like[2]: the reference's REAL*4 sums over 3300 lines carry ~1e-3 of noise, tests/test_hip_parity.py SGL_LONG_SUMS)
            compare(got, exp, rtol=2e-3 if (name, i) == ("sgl_real_like", 2) else 2e-4, what=f"sgl {name}[{i}]")


def test_fortran_shim_stops_like_the_reference(workdir, harness):
    """A missing TAPE3 is a STOP in the reference (src/lnfl_mod.f90:131-132): non-zero exit here."""
    g = Golden("cntnm_factors", workdir)
    case = os.path.join(workdir, "case_stop.bin")
    caseio.write_case(case, g.profiles[:1])
    r = subprocess.run([harness, case, os.path.join(workdir, "NO_SUCH_TAPE3"), os.path.join(workdir, "o.bin")],
                       cwd=workdir, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 or "HARNESS_SECONDS" not in r.stdout
    assert "ERROR OPENING HITRAN FILE" in (r.stdout + r.stderr)


def test_rtm_before_first_modm_still_loads_the_line_file(workdir, harness):
    """CALCTMR / RTM need no line file and may be called first; the reference then still loads TAPE3 in its first MODM
    call (INIT flag, src/modm.f90:187-190).  The shim creates a line-less context for the early calls and must replace it
    in MODM (round-1 advisor finding: it used to keep it and return zero line optical depths)."""
    g = Golden("c2_base", workdir)
    case = os.path.join(workdir, "case_rtmfirst.bin")
    out = os.path.join(workdir, "out_rtmfirst.bin")
    caseio.write_case(case, g.profiles)
    r = subprocess.run([harness, case, g.tape3, out], cwd=workdir, capture_output=True, text=True, timeout=300,
                       env={**os.environ, "HARNESS_RTM_FIRST": "1"})
    assert r.returncode == 0 and "HARNESS_SECONDS" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    for i, (got, exp) in enumerate(zip(caseio.read_dump(out), g.expected)):
        assert got.o_by_mol.max() > 0
        compare(got, exp, rtol=RTOL, what=f"rtm-first c2_base[{i}]")


def test_shim_writes_the_line_file_summary_to_the_log_unit(workdir, harness):
    """GET_LNFL / PRLNHD leave a "LINE FILE INFORMATION" block on unit IPR (MONORTM.LOG; src/lnfl_mod.f90:273-289).  The drop-in
    MODM writes the same block: compared line by line with the log of the compiled reference on the same TAPE3 (where the
    reference binary travelled with the snapshot), and against the header fields otherwise."""
    g = Golden("all_molecules", workdir)
    case = os.path.join(workdir, "case_log.bin")
    caseio.write_case(case, g.profiles[:1])
    run = os.path.join(workdir, "log_hip")
    os.makedirs(run, exist_ok=True)
    r = subprocess.run([harness, case, g.tape3, os.path.join(run, "o.bin")], cwd=run, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]

    def block(path):
        lines = open(path).read().splitlines()
        i = next(k for k, ln in enumerate(lines) if "LINE FILE INFORMATION" in ln)
        j = next(k for k, ln in enumerate(lines) if "TOTAL NUMBER OF LINES" in ln)
        return [ln.rstrip() for ln in lines[i:j + 1]]

    got = block(os.path.join(run, "HARNESS.LOG"))
    assert any("SUM LBLRTM" in ln for ln in got) and "LOWEST LINE" in got[-1]
    assert sum(" = " in ln for ln in got) >= 39 - 2          # one row per molecule of the file header
    ref = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "harness_ref_dbl")
    if os.path.exists(ref):
        rr = os.path.join(workdir, "log_ref")
        os.makedirs(rr, exist_ok=True)
        r2 = subprocess.run([ref, case, g.tape3, os.path.join(rr, "o.bin")], cwd=rr, capture_output=True, text=True, timeout=600)
        assert r2.returncode == 0, r2.stdout[-2000:] + r2.stderr[-2000:]
        assert got == block(os.path.join(rr, "HARNESS.LOG"))


def test_ragged_cross_section_files_packed_like_the_python_loader(workdir, harness):
    """A region whose first temperature file is SHORTER than its last one: the Fortran shim's packer
    (monortm_amd/fortran/xsec_hip.f90) must place every spectrum at a stride of the last file's point count, zero-filled, as
    monortm_amd/xsec.py flatten() does - the same tables, hence bit-identical optical depths through both routes."""
    from monortm_amd import api, xsec

    xd = os.path.join(workdir, "xs_ragged")
    names = xsec.synthetic_library(xd, f12_pres_mb=60.0, ragged=137)
    rng = np.random.default_rng(12)
    wn = np.sort(np.concatenate([rng.uniform(772.0, 811.0, 10), rng.uniform(831.0, 859.0, 8), [809.9, 810.8, 858.7]]))
    nlay = 5
    a = synth.standard_atmosphere(nlay, ztop_km=8)
    air = a["wbrodl"] / 0.781
    xamnt = np.stack([air * 1.0e-10, air * 2.6e-10, air * 5.3e-10], axis=1)
    pr = synth.Profile(wn=wn, p=a["p"], t=a["t"], tz=a["tz"], wkl=a["wkl"], wbrodl=a["wbrodl"], clw=a["clw"], irt=3, xs_names=names, xamnt=xamnt)
    pr.xs_dir = xd
    t3 = os.path.join(workdir, "TAPE3_ragged")
    tape3.write_tape3(t3, synth.synthetic_lines(40, seed=3, vlo=760.0, vhi=870.0))
    case, out = os.path.join(workdir, "case_ragged.bin"), os.path.join(workdir, "out_ragged.bin")
    caseio.write_case(case, [pr])
    r = subprocess.run([harness, case, t3, out], cwd=xd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "HARNESS_SECONDS" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    f = caseio.read_dump(out)[0]
    rt = api.MonoRTM(t3, wn[0], wn[-1])
    g = rt.run([pr])[0]
    rt.close()
    assert f.odxsec is not None and g.odxsec is not None and (g.odxsec > 0).any()
    assert np.array_equal(f.odxsec, g.odxsec) and np.array_equal(f.o, g.o)
