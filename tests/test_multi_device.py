"""Multi-GPU from the C ABI and the Fortran driver (north_star: "host code stays Fortran ... profiles shard across the 8
GPUs"; the shard axis is the independent-profile loop of src/monortm.f90:357).  A multi-device context shards a
host-buffer call into contiguous blocks of ceil(P/G) profiles; results must equal the one-device context bit for bit.
On the one-GPU test box MONORTM_DEVICES="0,0" / "0,0,0" puts the shards on the same card, which exercises the block
arithmetic (ragged and empty last blocks) exactly as G real devices would; with >= 2 visible devices the real ones are
used as well."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from common import ROOT
from monortm_amd import api, synth, tape3

pytestmark = pytest.mark.gpu


def _profiles(n, nwn=17):
    wn = synth.c2_channels(nwn, seed=8)
    return [synth.perturbed_profile(500 + i, wn, nlay=(20 if i % 3 else 13), cloud=(i % 2 == 0), irt=(1 if i % 2 else 3)) for i in range(n)]


def _same(a, b):
    for x, y in zip(a, b):
        for k in ("o", "o_by_mol", "oc", "o_clw", "rup", "rdn", "trtot", "rad", "tb", "tmr"):
            assert np.array_equal(getattr(x, k), getattr(y, k)), k
        assert x.tmpsfc_out == y.tmpsfc_out


@pytest.mark.parametrize("devices,nprof,real_kind", [("0,0", 5, 8), ("0,0,0", 7, 8), ("0,0", 1, 8), ("0,0,0,0", 3, 4)])
def test_multi_context_equals_single(tmp_path, monkeypatch, devices, nprof, real_kind):
    t3 = str(tmp_path / "TAPE3")
    tape3.write_tape3(t3, synth.synthetic_lines(150, seed=12, lc_frac=0.5, sdep_frac=0.1))
    profs = _profiles(nprof)
    one = api.MonoRTM(t3, profs[0].wn[0], profs[0].wn[-1], real_kind=real_kind)
    want = one.run(profs)
    one.close()
    monkeypatch.setenv("MONORTM_DEVICES", devices)
    multi = api.MonoRTM(t3, profs[0].wn[0], profs[0].wn[-1], real_kind=real_kind, ngpu=0)
    assert multi.lib.monortm_hip_device_count(multi.ctx) == len(devices.split(","))
    assert multi.line_count(0) == 150
    got = multi.run(profs)
    _same(got, want)
    # the per-device resident O is found again by the sharded RTM call
    assert multi.lib.monortm_hip_counter(multi.ctx, 0) == min(nprof, len(devices.split(",")))
    # device pointers / timers belong to one device
    assert multi.lib.monortm_hip_profile(multi.ctx, 1) == 6
    multi.close()


def test_real_devices_when_present(tmp_path):
    import torch

    g = min(2, torch.cuda.device_count())
    t3 = str(tmp_path / "TAPE3")
    tape3.write_tape3(t3, synth.synthetic_lines(150, seed=12))
    profs = _profiles(6)
    one = api.MonoRTM(t3, profs[0].wn[0], profs[0].wn[-1])
    want = one.run(profs)
    one.close()
    multi = api.MonoRTM(t3, profs[0].wn[0], profs[0].wn[-1], ngpu=g)
    assert multi.lib.monortm_hip_device_count(multi.ctx) == g
    _same(multi.run(profs), want)
    multi.close()
    with pytest.raises(api.MonoRTMError):  # more devices than the node has
        api.MonoRTM(t3, profs[0].wn[0], profs[0].wn[-1], ngpu=torch.cuda.device_count() + 1)


def test_sharded_calls_leave_the_current_device_alone(tmp_path):
    """monortm_hip_init_multi and the sharded host-buffer calls select every shard's device in turn (hipSetDevice is per
    thread): the caller's current device must be the same afterwards - a torch allocation or a *_dev call on a context of
    device 0 made next would otherwise land on the last shard's device.  Meaningful with >= 2 GPUs; on one GPU it
    still runs the code path (device 0 throughout)."""
    import torch

    t3 = str(tmp_path / "TAPE3")
    tape3.write_tape3(t3, synth.synthetic_lines(150, seed=12))
    profs = _profiles(5)
    g = max(1, min(2, torch.cuda.device_count()))
    for cur in range(g):
        torch.cuda.set_device(cur)
        multi = api.MonoRTM(t3, profs[0].wn[0], profs[0].wn[-1], ngpu=g)
        assert torch.cuda.current_device() == cur
        multi.run(profs)
        assert torch.cuda.current_device() == cur
        multi.close()
        assert torch.cuda.current_device() == cur
        t = torch.zeros(4, device="cuda")
        assert t.device.index == cur
    torch.cuda.set_device(0)


def test_fortran_driver_shards_profiles(tmp_path):
    """MONORTM_NGPU in the own Fortran driver: the 3-profile IATM=0 deck over 2 device contexts gives the same MONORTM.OUT
    as over one."""
    from monortm_amd import _build

    exe = _build.build_fortran_shim()["driver"]
    decks = os.path.join(ROOT, "tests", "golden", "decks")
    case = "case45_IATM0_three_profiles"
    outs = []
    for tag, env in (("one", {}), ("two", {"MONORTM_NGPU": "2", "MONORTM_DEVICES": "0,0"})):
        d = tmp_path / tag
        d.mkdir()
        for f in os.listdir(os.path.join(decks, case)):
            if f.endswith(".IN"):
                shutil.copy(os.path.join(decks, case, f), d)
        shutil.copy(os.path.join(decks, "TAPE3_synthetic"), d / "TAPE3")
        r = subprocess.run([exe], cwd=d, capture_output=True, text=True, timeout=600, env={**os.environ, **env})
        assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
        if env:
            assert "sharded over  2 device" in r.stdout
        outs.append(open(d / "MONORTM.OUT").read())
    assert outs[0] == outs[1] and len(outs[0]) > 1000
