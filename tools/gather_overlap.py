"""Does the output gather of a profile-sharded job find wave slots behind lines_kernel?  (GPU box, repo root)

    python tools/gather_overlap.py > gpurun_out/r05_gather_overlap.txt

On ONE GPU, with the 128-profile share of configs[3] that each of 8 GPUs holds (c4shard: lines_kernel fills every one of the
4096 wave slots that 128 VGPRs leave on 256 CUs, and raises its waves to s_setprio 1-3 when a.fair is on), a communication
kernel is issued on a SECOND stream while the step's kernels run on the first, exactly as distributed.GatherPlan does it
(copy of the step's outputs, then the collective, asynchronously, waited for one step later):
  * "rccl":  monortm_hip_gather_dev with a world of 1 = librccl's own ncclGather kernel on this GPU (the transport is a
             local copy, the kernel launch, its wave-slot needs and its priority are the real ones);
  * "copy":  a plain device-to-device copy of the same 307 KB on the side stream (what torch's gloo / nccl path adds around it).
Reported per setting of `fair` (wave priorities on / off / auto): step time alone, step time with the concurrent gather,
the gather's own latency alone and while the kernels run (HIP events on the side stream).  If the communication kernel had
to wait for the grid to drain, its concurrent latency would approach a step (0.18 ms) and the 8-GPU strong-scaling step
would pay it; if it finds slots, the latency stays near the idle one and the step time barely moves.
"""
from __future__ import annotations

import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch

    from monortm_amd import api, synth, tape3

    torch.cuda.set_device(0)
    tmp = tempfile.mkdtemp(prefix="gather_overlap_")
    t3 = os.path.join(tmp, "TAPE3")
    tape3.write_tape3(t3, synth.synthetic_lines(500))
    wn = synth.c2_channels(50)
    profs = [synth.perturbed_profile(i, wn, nlay=64) for i in range(128)]
    rt = api.MonoRTM(t3, wn[0], wn[-1], device=0)
    b = api.DeviceBatch(rt, profs, device="cuda:0")
    rt.comm_init(1, 0, api.MonoRTM.comm_unique_id())
    side = torch.cuda.Stream()
    main_s = torch.cuda.current_stream()
    out0 = b.spectral_block()
    send = torch.empty_like(out0)
    recv = torch.empty_like(out0)
    send4 = torch.empty((128, 6, 50), dtype=out0.dtype, device=out0.device)   # round 4's row-major copy
    print(f"# gather payload: {send.numel() * send.element_size()} bytes ([128, 6, 50] f64), step = lines + finish + rtm kernels of c4shard")

    def gather(kind):
        if kind == "rccl":
            rt.gather_dev(send, recv, 0, side.cuda_stream)
        else:
            with torch.cuda.stream(side):
                recv.view(-1).copy_(send.reshape(-1), non_blocking=True)

    block = [False]   # round 5: the kernels' own output block is gathered (two blocks in turn), no stack / copy on the compute stream

    def run(kind, steps, concurrent):
        """-> (ms per step, mean gather latency in us).  concurrent: the gather of step k travels while step k + 1 runs."""
        nonlocal send
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        b.pingpong = block[0]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            if concurrent or kind is None:
                b.step()
            if kind is not None:
                if concurrent:
                    if block[0]:
                        main_s.wait_stream(side)                              # GatherPlan.start(): the gather before the last one is done ...
                        send = b.spectral_block()                             # ... and this step's block goes as it is
                    else:
                        send4.copy_(torch.stack([b.RAD, b.TB, b.TRTOT, b.TMR, b.RUP, b.RDN], dim=1), non_blocking=True)   # round 4: a stacked copy ...
                        send = send4
                    side.wait_stream(main_s)                                  # ... and the collective behind it on its own stream
                ev[k][0].record(side)
                gather(kind)
                ev[k][1].record(side)
                if not concurrent:
                    side.synchronize()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps * 1e3
        lat = np.mean([a.elapsed_time(c) for a, c in ev]) * 1e3 if kind is not None else float("nan")
        return dt, lat

    lo_side, hi_side = side, torch.cuda.Stream(priority=-1)
    for fair, blk, hi in (("auto", False, False), ("auto", True, False), ("auto", True, True), ("0", True, True)):
        rt.set_option("fair", fair)
        block[0] = blk
        side = hi_side if hi else lo_side
        print(f"--- fair={fair}, gather source: " + ("the kernels' output block, ping-pong (round 5)" if blk else "torch.stack + copy of the six outputs (round 4)")
              + (", gather on a HIGH-PRIORITY stream" if hi else ", gather on a default-priority stream"))
        for _ in range(300):
            b.step()
        torch.cuda.synchronize()
        alone, _ = run(None, 400, True)
        alone = min(alone, run(None, 400, True)[0])
        print(f"fair={fair}: step alone {alone:.4f} ms")
        for kind in ("rccl", "copy"):
            run(kind, 50, False)
            _, lat_idle = run(kind, 200, False)
            run(kind, 50, True)
            dt, lat_busy = run(kind, 400, True)
            dt2, lat_busy2 = run(kind, 400, True)
            if dt2 < dt:
                dt, lat_busy = dt2, lat_busy2
            print(f"fair={fair} {kind:4s}: gather alone {lat_idle:7.1f} us | concurrent with the next step: gather {lat_busy:7.1f} us, "
                  f"step {dt:.4f} ms ({(dt / alone - 1) * 100:+.1f} % against the step alone)")
    b.check()
    rt.close()


if __name__ == "__main__":
    main()
