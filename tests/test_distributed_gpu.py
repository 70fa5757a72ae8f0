"""N > 1 with the HIP path as the per-block compute: two ranks share the one card of the test box (gloo rendezvous;
RCCL refuses two ranks on one device), each runs DeviceBatch.step on its block of profiles and the spectral outputs
are gathered to rank 0 by distributed.run_sharded - the production sharding driver.  Float64 and float32 contexts,
an empty last block included.  Result must equal the single-process batch bit for bit (profiles are independent)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from common import ROOT

pytestmark = pytest.mark.gpu


def _case(nprof):
    from monortm_amd import synth

    wn = synth.c2_channels(20, seed=5)
    return wn, [synth.perturbed_profile(300 + i, wn, nlay=24, cloud=(i % 2 == 0), irt=(1 if i % 2 else 3)) for i in range(nprof)]


def _spectral(t3, profs, real_kind):
    from monortm_amd import api

    rt = api.MonoRTM(t3, profs[0].wn[0], profs[0].wn[-1], device=0, real_kind=real_kind)
    b = api.DeviceBatch(rt, profs, device="cuda:0")
    b.step()
    b.check()
    out = b.spectral_outputs().clone()
    torch.cuda.synchronize()
    rt.close()
    return out


def _worker(rank, world, port, t3, nprof, real_kind, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from monortm_amd import distributed as D

    _, profs = _case(nprof)
    out = D.run_sharded(profs, lambda block: _spectral(t3, block, real_kind))
    if rank == 0:
        q.put(out.cpu().numpy())
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("nprof,real_kind", [(5, 8), (1, 4)])
def test_run_sharded_with_hip_compute(workdir, nprof, real_kind):
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests need the MI355X")
    from monortm_amd import synth, tape3

    t3 = os.path.join(workdir, "TAPE3_dist_gpu")
    tape3.write_tape3(t3, synth.synthetic_lines(120, seed=11, lc_frac=0.5))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, t3, nprof, real_kind, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    _, profs = _case(nprof)
    want = _spectral(t3, profs, real_kind).cpu().numpy()
    assert got.dtype == want.dtype and got.shape == want.shape == (nprof, 6, 20)
    assert np.array_equal(got, want)
