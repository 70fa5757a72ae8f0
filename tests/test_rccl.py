"""RCCL's first run must not be the driver's scaling bench: where at least two GPUs are visible, (i) the
production sharding driver distributed.run_sharded runs on the "nccl" backend - one rank per GPU, HIP compute per block, ONE
gather to rank 0 - and must equal the single-process batch bit for bit; (ii) `python bench.py --gpus 2` (its own launcher,
nccl) must report two ranks on two distinct devices.  On a one-GPU box both skip with the reason (RCCL refuses two ranks
on one device; tests/test_distributed_gpu.py covers that box with a gloo rendezvous and the same HIP compute)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from common import ROOT
from test_distributed_gpu import _case

pytestmark = pytest.mark.gpu


def _need_two_gpus():
    n = torch.cuda.device_count()   # (does not initialise the GPU on this image)
    if n < 2:
        pytest.skip(f"{n} GPU visible: the RCCL path needs one device per rank (at least 2)")
    return n


def _spectral_on(t3, profs, real_kind, device):
    from monortm_amd import api

    torch.cuda.set_device(device)
    rt = api.MonoRTM(t3, profs[0].wn[0], profs[0].wn[-1], device=device, real_kind=real_kind)
    b = api.DeviceBatch(rt, profs, device=f"cuda:{device}")
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
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{rank}"))
    from monortm_amd import distributed as D

    _, profs = _case(nprof)
    out = D.run_sharded(profs, lambda block: _spectral_on(t3, block, real_kind, rank))
    if rank == 0:
        q.put(out.cpu().numpy())
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("nprof,real_kind", [(7, 8), (1, 4)])
def test_run_sharded_over_rccl(workdir, nprof, real_kind):
    _need_two_gpus()
    from monortm_amd import synth, tape3

    t3 = os.path.join(workdir, "TAPE3_rccl")
    tape3.write_tape3(t3, synth.synthetic_lines(120, seed=11, lc_frac=0.5))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")   # fresh children: nothing here has touched the GPU yet
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, t3, nprof, real_kind, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=600)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    _, profs = _case(nprof)
    want = _spectral_on(t3, profs, real_kind, 0).cpu().numpy()
    assert got.dtype == want.dtype and got.shape == want.shape == (nprof, 6, 20)
    assert np.array_equal(got, want)      # profiles are independent: sharding changes no bit


def test_bench_two_ranks_over_rccl():
    _need_two_gpus()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--min-seconds", "0.2",
                        "--no-extra", "--no-pmc", "--no-cpu-baseline"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["n_ranks_seen"] == 2
    devs = {d[1] for d in out["rank_devices"]}
    assert len(devs) == 2, out["rank_devices"]
    assert out["scaling"] == "strong" and out["value"] > 0   # configs[3]: 1024 profiles split over the ranks
    assert "one RCCL gather/step" in out["config"]["parallelism"]


# ---- the C-ABI gather (monortm_hip_comm_init / monortm_hip_gather_dev): what a C / Fortran job uses instead of torch.distributed
def test_gather_dev_single_rank_runs_rccl_on_this_gpu(workdir):
    """world = 1: RCCL is loaded (dlopen), a communicator is built on the context's device and ncclGather runs on the GPU that
    is here - the entry points are exercised on hardware even where only one device is visible."""
    from monortm_amd import api

    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests need the MI355X")
    rt = api.MonoRTM("", 0.0, 0.0)
    try:
        rt.comm_init(1, 0, api.MonoRTM.comm_unique_id())
        send = torch.arange(6 * 50, dtype=torch.float64, device="cuda:0").reshape(6, 50) * 0.5
        recv = torch.zeros_like(send)
        rt.gather_dev(send, recv, root=0, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert torch.equal(send, recv)
        with pytest.raises(api.MonoRTMError):
            rt.gather_dev(send, recv, root=3)       # no such rank
    finally:
        rt.close()


def _gather_worker(rank, world, uid, q):
    sys.path.insert(0, ROOT)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from monortm_amd import api

    torch.cuda.set_device(rank)
    rt = api.MonoRTM("", 0.0, 0.0, device=rank)
    rt.comm_init(world, rank, uid)
    send = torch.full((3, 6, 20), float(rank + 1), dtype=torch.float64, device=f"cuda:{rank}")
    recv = torch.zeros((world, 3, 6, 20), dtype=torch.float64, device=f"cuda:{rank}") if rank == 0 else None
    rt.gather_dev(send, recv, root=0, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    if rank == 0:
        q.put(recv.cpu().numpy())
    rt.close()


def test_gather_dev_two_ranks():
    _need_two_gpus()
    from monortm_amd import api

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    uid = api.MonoRTM.comm_unique_id()   # (ncclGetUniqueId touches no device)
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, uid, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=600)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    assert got.shape == (2, 3, 6, 20) and (got[0] == 1.0).all() and (got[1] == 2.0).all()
