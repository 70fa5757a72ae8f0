"""N > 1 path on CPU: world_size-2 gloo processes shard the profiles, compute their block and gather to
rank 0.  There is no GPU here, so the per-block compute injected into the sharding driver is the CPU
oracle (test infrastructure); on the GPU box the same driver is fed by the HIP path (bench.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from common import ROOT
from monortm_amd import distributed as D


def test_shard_bounds():
    assert D.shard_bounds(1024, 8) == [(i * 128, (i + 1) * 128) for i in range(8)]
    assert D.shard_bounds(5, 2) == [(0, 3), (3, 5)]
    assert D.shard_bounds(3, 4) == [(0, 1), (1, 2), (2, 3), (3, 3)]
    b = D.shard_bounds(1000, 7)
    assert b[0][0] == 0 and b[-1][1] == 1000 and all(x[1] == y[0] for x, y in zip(b, b[1:]))


def _worker(rank, world, port, t3, nprof, q, f32=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from monortm_amd import synth
    from oracle.pyoracle import Oracle

    wn = synth.c2_channels(6, seed=2)
    profs = [synth.perturbed_profile(i, wn, nlay=8, cloud=(i % 2 == 1)) for i in range(nprof)]
    orc = Oracle(t3, wn[0], wn[-1])

    def compute(block):
        rows = []
        for p in block:
            d = orc.run(p)
            rows.append(np.stack([d.rad, d.tb, d.trtot, d.tmr, d.rup, d.rdn]))
        out = torch.from_numpy(np.stack(rows))
        return out.float() if f32 else out  # the sgl build's blocks are float32: the empty rank must follow

    out = D.run_sharded(profs, compute)
    if rank == 0:
        q.put(out.numpy())
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,nprof,f32", [(2, 5, False), (2, 1, False), (2, 1, True)])
def test_profile_sharding_gather_gloo(workdir, world, nprof, f32):
    from monortm_amd import synth, tape3
    from oracle.pyoracle import Oracle, build

    build()
    t3 = os.path.join(workdir, "TAPE3_dist")
    tape3.write_tape3(t3, synth.synthetic_lines(40, seed=3))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, t3, nprof, q, f32)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=180)
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    # serial result
    wn = synth.c2_channels(6, seed=2)
    orc = Oracle(t3, wn[0], wn[-1])
    for i in range(nprof):
        d = orc.run(synth.perturbed_profile(i, wn, nlay=8, cloud=(i % 2 == 1)))
        exp = np.stack([d.rad, d.tb, d.trtot, d.tmr, d.rup, d.rdn])
        assert np.array_equal(got[i], exp.astype(np.float32) if f32 else exp)
    assert got.shape == (nprof, 6, 6) and got.dtype == (np.float32 if f32 else np.float64)


def _plan_worker(rank, world, port, nprof, q, field_major=False):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = D.shard_bounds(nprof, world)[rank]
    like = torch.zeros((hi - lo, 6, 4), dtype=torch.float64)
    if field_major:   # the kernels' own output block [6, n_local, nwn], two of them written in turn (api.DeviceBatch.pingpong)
        like = like.permute(1, 0, 2).contiguous()
    plan = D.GatherPlan(nprof, like, field_major=field_major)
    blocks = [torch.zeros_like(like), torch.zeros_like(like)]
    outs = []
    for step in range(3):  # the bench loop: start the gather of step k, compute step k+1 meanwhile
        local = torch.stack([torch.full((6, 4), float(1000 * step + i)) for i in range(lo, hi)]).double() if hi > lo else like
        if field_major:
            if hi > lo:
                blocks[step & 1].copy_(local.permute(1, 0, 2))
            local = blocks[step & 1]
        plan.start(local)
        res = plan.result()
        if rank == 0:
            outs.append(res.clone().numpy())
    if rank == 0:
        q.put(outs)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,nprof,field_major", [(2, 5, False), (2, 4, False), (2, 5, True), (2, 4, True), (3, 4, True)])
def test_gather_plan_gloo(world, nprof, field_major):
    """GatherPlan (what bench.py uses for N > 1): preallocated receive slots, asynchronous issue, ragged last block; row-major
    rows and the field-major block that bench.py hands over without a copy (round 5)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_plan_worker, args=(r, world, port, nprof, q, field_major)) for r in range(world)]
    for p in procs:
        p.start()
    outs = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for step, o in enumerate(outs):
        assert o.shape == (nprof, 6, 4)
        assert np.array_equal(o[:, 0, 0], 1000.0 * step + np.arange(nprof))
