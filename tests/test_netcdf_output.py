"""netCDF output of the file interface (reference: STOREOUT's USENETCDF branch, src/monortm_sub.F90:698-778).  The writer is
own Fortran (monortm_amd/fortran/netcdf3_writer.f90, classic CDF-1 format, no libnetcdff); a netCDF library reads it back."""
import os
import subprocess

import numpy as np
import pytest
from scipy.io import netcdf_file

from common import ROOT

FC = "/opt/rocm/bin/amdflang"
NAMES = ["FREQUENCY", "BT", "RAD", "TRANS", "PWV", "CLW", "SFCT", "EMIS", "REFL", "ANGLE", "TMR", "TOTAL_OD", "TOTAL_OD_BY_MOLECULE",
         "XSEC_OD", "MOLECULE", "LAYER_OPTICAL_DEPTH", "LAYER_OPTICAL_DEPTH_BY_MOLECULE"]


def _check_layout(nc, nwn, kount, nlay):
    assert dict(nc.dimensions) == {"FREQUENCY": nwn, "MOLECULE": kount, "LAYERS": nlay, "STRING_LENGTH": 8}
    assert list(nc.variables) == NAMES      # the reference's seventeen variables, in its order of definition
    v = nc.variables
    for n in NAMES[:12] + ["XSEC_OD"]:
        assert v[n].dimensions == ("FREQUENCY",) and v[n].data.dtype == np.dtype(">f8"), n
    assert v["TOTAL_OD_BY_MOLECULE"].dimensions == ("FREQUENCY", "MOLECULE")          # Fortran dimids (mol, wn)
    assert v["MOLECULE"].dimensions == ("MOLECULE", "STRING_LENGTH")
    assert v["LAYER_OPTICAL_DEPTH"].dimensions == ("LAYERS", "FREQUENCY")
    assert v["LAYER_OPTICAL_DEPTH_BY_MOLECULE"].dimensions == ("LAYERS", "MOLECULE", "FREQUENCY")
    assert v["LAYER_OPTICAL_DEPTH_BY_MOLECULE"].data.dtype == np.dtype(">f4")         # NF90_FLOAT in every build (:744)
    assert v["FREQUENCY"].units.decode().startswith("FREQ(") and len(v["FREQUENCY"].units) == 11   # NF_PUT_ATT_TEXT(.., 11, wnunits)


def test_writer_round_trip(workdir):
    exe = os.path.join(workdir, "nc_check")
    subprocess.check_call([FC, "-O1", "-module-dir", workdir, os.path.join(ROOT, "monortm_amd", "fortran", "netcdf3_writer.f90"),
                           os.path.join(ROOT, "examples", "nc_writer_check.f90"), "-o", exe], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    subprocess.check_call([exe], cwd=workdir)
    nc = netcdf_file(os.path.join(workdir, "MONORTM.00001.nc"), "r", mmap=False)
    nwn, kount, nlay = 5, 3, 4
    _check_layout(nc, nwn, kount, nlay)
    v = nc.variables
    i = np.arange(1, nwn + 1.0)
    assert np.array_equal(v["FREQUENCY"].data, 22.0 + i) and np.array_equal(v["BT"].data, 250.0 + 0.5 * i)
    assert np.allclose(v["RAD"].data, 1e-7 * i, rtol=1e-15) and np.allclose(v["TRANS"].data, 0.9 - 0.01 * i, rtol=1e-15)
    for n, c in (("PWV", 1.25), ("CLW", 0.03), ("SFCT", 2.75), ("EMIS", 0.6), ("REFL", 0.4), ("ANGLE", 180.0)):
        assert np.allclose(v[n].data, c, rtol=1e-15), n
    assert np.array_equal(v["TMR"].data, 270.0 + i) and np.allclose(v["TOTAL_OD"].data, 0.1 * i) and np.allclose(v["XSEC_OD"].data, 1e-3 * i)
    k = np.arange(1, kount + 1.0)
    assert np.allclose(v["TOTAL_OD_BY_MOLECULE"].data, k[None, :] + 0.01 * i[:, None])
    assert [b"".join(r).decode() for r in v["MOLECULE"].data] == ["  H2O   ", "  CO2   ", "   O2   "]
    j = np.arange(1, nlay + 1.0)
    assert np.array_equal(v["LAYER_OPTICAL_DEPTH"].data, i[None, :] + 0.25 * j[:, None])
    assert np.array_equal(v["LAYER_OPTICAL_DEPTH_BY_MOLECULE"].data, (i[None, None, :] + 10.0 * k[None, :, None] + 100.0 * j[:, None, None]).astype(np.float32))
    nc.close()


@pytest.mark.gpu
def test_driver_writes_netcdf_matching_monortm_out(workdir):
    """The own driver on the three-profile IATM = 0 deck with MONORTM_NETCDF=1: one MONORTM.NNNNN.nc per profile whose spectral
    variables equal the columns of MONORTM.OUT (to its printed precision) and whose layer optical depths add up to TOTAL_OD."""
    import shutil

    from test_reference_driver_dropin import DECKS, stage_inputs

    run = os.path.join(workdir, "nc_run")
    os.makedirs(run, exist_ok=True)
    stage_inputs(os.path.join(DECKS, "case45_IATM0_three_profiles"), run)
    shutil.copy(os.path.join(DECKS, "TAPE3_synthetic"), os.path.join(run, "TAPE3"))
    env = dict(os.environ, MONORTM_NETCDF="1")
    drv = os.path.join(ROOT, "monortm_amd", "lib", "monortm_hip")
    r = subprocess.run([drv], cwd=run, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert sorted(f for f in os.listdir(run) if f.endswith(".nc")) == ["MONORTM.00001.nc", "MONORTM.00002.nc", "MONORTM.00003.nc"]
    rows = [ln.split() for ln in open(os.path.join(run, "MONORTM.OUT")) if ln[:5].strip().isdigit()]
    prof1 = [x for x in rows if int(x[0]) == 2]
    nc = netcdf_file(os.path.join(run, "MONORTM.00002.nc"), "r", mmap=False)
    v = nc.variables
    nwn = len(prof1)
    _check_layout(nc, nwn, nc.dimensions["MOLECULE"], nc.dimensions["LAYERS"])
    assert np.allclose(v["FREQUENCY"].data, [float(x[1]) for x in prof1], atol=6e-4)
    assert np.allclose(v["BT"].data, [float(x[2]) for x in prof1], atol=6e-6)
    assert np.allclose(v["RAD"].data, [float(x[4]) for x in prof1], rtol=2e-9)
    assert np.allclose(v["TOTAL_OD"].data, [float(x[12]) for x in prof1], rtol=2e-4)
    assert np.allclose(v["LAYER_OPTICAL_DEPTH"].data.sum(axis=0), v["TOTAL_OD"].data, rtol=1e-12)
    nc.close()
    shutil.rmtree(run, ignore_errors=True)
