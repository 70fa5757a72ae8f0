"""CPU-only checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports every
symbol include/monortm_hip.h declares (no compute calls - there is no GPU here)."""
import ctypes
import os
import re

import pytest

from common import ROOT
from monortm_amd import _build, api


def declared_functions():
    hdr = open(os.path.join(ROOT, "include", "monortm_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(monortm_hip_\w+)\s*\(", hdr)))


def test_header_and_binding_agree():
    assert declared_functions() == sorted(api.SYMBOLS)


def test_library_builds_and_exports_every_symbol():
    so = _build.build_hip()
    lib = ctypes.CDLL(so)
    for name in declared_functions():
        assert hasattr(lib, name), f"{name} declared in include/monortm_hip.h but not exported by {so}"
    api.load_library()


def test_no_gpu_means_loud_failure(tmp_path):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from monortm_amd import synth, tape3

    t3 = str(tmp_path / "TAPE3")
    tape3.write_tape3(t3, synth.synthetic_lines(20))
    with pytest.raises(api.MonoRTMError):
        api.MonoRTM(t3, 0.3, 30.0)


def test_product_does_not_import_oracle():
    """The shipped package must never route through the CPU oracle."""
    pkg = os.path.join(ROOT, "monortm_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".f90", ".h")):
                src = open(os.path.join(dp, f), errors="replace").read()
                assert "pyoracle" not in src and "liboracle" not in src and "monortm_oracle" not in src, f


def test_null_context_is_an_error_not_a_crash():
    """Every entry point that takes a context returns MONORTM_EARG for NULL (no GPU is touched)."""
    lib = api.load_library()
    ms, n = ctypes.c_double(), ctypes.c_longlong()
    assert lib.monortm_hip_profile(None, 1) == 6
    assert lib.monortm_hip_kernel_time(None, 0, ctypes.byref(ms), ctypes.byref(n)) == 6
    assert lib.monortm_hip_check(None, None) == 6
    assert lib.monortm_hip_has_lines(None) == 0
    assert lib.monortm_hip_xsec_regions(None) == 0
    assert lib.monortm_hip_line_count(None, 0) == -1
    z = [None] * 30
    assert lib.monortm_hip_modm_dev(None, 1, 1, None, 0.0, None, 1, 7, *z[:6], 1.0, 1.0, 0.0, 0, 0, *z[:6]) == 6
    assert lib.monortm_hip_rtm_dev(None, 1, 1, None, None, 1, None, 1, *z[:13]) == 6
    assert lib.monortm_hip_modm(None, 1, 1, None, 0.0, None, 1, 7, *z[:6], 1.0, 1.0, 0.0, 0, 0, *z[:4]) == 6
    assert lib.monortm_hip_rtm(None, 1, 1, None, None, 1, None, 1, *z[:12]) == 6
    assert b"null context" in lib.monortm_hip_last_error(None)
    lib.monortm_hip_finalize(None)


@pytest.mark.gpu
def test_device_argument_guards(tmp_path):
    """A context without a TAPE3 refuses MODM; nlay[p] > nlay_max and descending wavenumbers handed over in DEVICE
    memory are flagged by the kernels and reported by monortm_hip_check (the host-buffer calls validate on the host)."""
    import numpy as np
    import torch

    from monortm_amd import synth, tape3

    wn = synth.c2_channels(8, seed=3)
    profs = [synth.perturbed_profile(i, wn, nlay=6) for i in range(2)]
    rt0 = api.MonoRTM("", 0.0, 0.0)
    assert rt0.lib.monortm_hip_has_lines(rt0.ctx) == 0
    with pytest.raises(api.MonoRTMError) as e:
        rt0.modm(profs)
    assert e.value.code == 6
    rt0.close()

    t3 = str(tmp_path / "TAPE3")
    tape3.write_tape3(t3, synth.synthetic_lines(60, seed=2))
    rt = api.MonoRTM(t3, wn[0], wn[-1])
    assert rt.lib.monortm_hip_has_lines(rt.ctx) == 1
    b = api.DeviceBatch(rt, profs)
    b.step()
    b.check()  # clean
    b.nlay[1] = 7  # > nlay_max = 6, in device memory
    b.step()
    with pytest.raises(api.MonoRTMError) as e:
        b.check()
    assert e.value.code == 6
    b.nlay[1] = 6
    b.step()
    b.check()  # the flag was cleared
    b.wn.copy_(torch.as_tensor(np.ascontiguousarray(wn[::-1])))
    b.step()
    with pytest.raises(api.MonoRTMError) as e:
        b.check()
    assert e.value.code == 6
    with pytest.raises(api.MonoRTMError):  # host-buffer call: validated before anything is launched
        p = synth.perturbed_profile(0, wn[::-1].copy(), nlay=6)
        rt.modm([p])
    rt.close()
    # DVSET /= 0 promises the grid V1 + i DVSET: a device-resident grid that breaks the promise is flagged as well
    import dataclasses

    wg = 2.0 + 0.01 * np.arange(40)
    pg = [dataclasses.replace(synth.perturbed_profile(0, wg, nlay=6), dvset=0.01)]
    rt = api.MonoRTM(t3, wg[0], wg[-1])
    b = api.DeviceBatch(rt, pg)
    b.step()
    b.check()  # uniform: clean
    wbad = wg.copy()
    wbad[17] += 0.003
    b.wn.copy_(torch.as_tensor(wbad))
    b.step()
    with pytest.raises(api.MonoRTMError) as e:
        b.check()
    assert e.value.code == 6
    # the grid of the reference's "sgl" driver - WN(J) = V1 + (J-1)*DVSET with a REAL*4 product (src/monortm_sub.F90:287):
    # consecutive differences are off by up to 6e-8 J DVSET, the points stay within a fraction of a step: accepted
    dv4 = np.float32(0.01)
    wsgl = 2.0 + (np.arange(40, dtype=np.float32) * dv4).astype(np.float64)
    assert np.max(np.abs(np.diff(wsgl) - float(dv4))) > 1e-6 * float(dv4)
    b.wn.copy_(torch.as_tensor(wsgl))
    b.step()
    b.check()
    rt.close()
    rt = api.MonoRTM(t3, wsgl[0], wsgl[-1])
    rt.modm([dataclasses.replace(synth.perturbed_profile(0, wsgl, nlay=6), dvset=float(dv4))])   # host-buffer route: accepted too
    with pytest.raises(api.MonoRTMError):
        rt.modm([dataclasses.replace(synth.perturbed_profile(0, wbad, nlay=6), dvset=0.01)])
    rt.close()


@pytest.mark.gpu
def test_set_option_parses_strictly(tmp_path):
    """monortm_hip_set_option refuses what it cannot parse ('nslice'='abc' used to become 1, 'fair'='x' 0), and a
    mistyped MONORTM_* switch in the environment fails the init instead of running silently with the default."""
    import os
    import subprocess
    import sys

    import numpy as np

    from monortm_amd import synth, tape3

    wn = synth.c2_channels(5, seed=2)
    t3 = str(tmp_path / "TAPE3")
    tape3.write_tape3(t3, synth.synthetic_lines(30, seed=2))
    rt = api.MonoRTM(t3, wn[0], wn[-1])
    for name, good in (("nslice", ("auto", "1", "16")), ("fair", ("auto", "0", "1")), ("tile_waves", ("auto", "1", "2", "4")), ("lines_kernel", ("auto", "wn"))):
        for v in good:
            rt.set_option(name, v)
    for name, bad in (("nslice", ("abc", "0", "17", "4x", "-1")), ("fair", ("x", "2", "1.5")), ("tile_waves", ("3", "8", "two")),
                      ("lines_kernel", ("state", "p", "w", "wnx")), ("no_such_option", ("1",))):
        for v in bad:
            with pytest.raises(api.MonoRTMError) as e:
                rt.set_option(name, v)
            assert e.value.code == 6, (name, v)
    rt.close()
    code = ("import sys; sys.path.insert(0, %r)\nfrom monortm_amd import api\n"
            "try:\n    api.MonoRTM(%r, %r, %r)\nexcept api.MonoRTMError as e:\n    print('REFUSED', e.code, e)\n" % (ROOT, t3, float(wn[0]), float(wn[-1])))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, MONORTM_NSLICE="abc"), capture_output=True, text=True, timeout=300)
    assert "REFUSED 6" in r.stdout and "MONORTM_NSLICE" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_rtm_reuses_resident_optical_depths(tmp_path):
    """CALCTMR / RTM called with exactly the O that MODM returned (the reference driver's sequence) read it from device
    memory; any other O is uploaded.  Both routes give the same numbers."""
    import numpy as np

    from monortm_amd import synth, tape3

    wn = synth.c2_channels(9, seed=4)
    profs = [synth.perturbed_profile(i, wn, nlay=12, irt=(1 if i else 3)) for i in range(2)]
    t3 = str(tmp_path / "TAPE3")
    tape3.write_tape3(t3, synth.synthetic_lines(80, seed=6))
    rt = api.MonoRTM(t3, wn[0], wn[-1])
    cnt = lambda: rt.lib.monortm_hip_counter(rt.ctx, 0)  # noqa: E731
    O = rt.modm(profs)[0]
    a = rt.rtm(profs, O)
    assert cnt() == 1
    b = rt.rtm(profs, O.copy())          # same values at another address: still resident
    assert cnt() == 2
    O2 = O.copy()
    O2[0, 0, 0] *= 1.5
    c = rt.rtm(profs, O2)                # different values: uploaded
    assert cnt() == 2
    assert all(np.array_equal(x, y) for x, y in zip(a[:6], b[:6]))
    assert not np.array_equal(a[3], c[3])
    rt2 = api.MonoRTM(t3, wn[0], wn[-1])  # a context that never ran MODM takes the upload route for the same O
    d = rt2.rtm(profs, O)
    assert rt2.lib.monortm_hip_counter(rt2.ctx, 0) == 0
    assert all(np.array_equal(x, y) for x, y in zip(a[:6], d[:6]))
    rt.close()
    rt2.close()
