#!/usr/bin/env python3
"""lines_ms_kernel against lines_kernel for channel sets of other sizes than configs[3]'s 50 (GPU box, repo root):
    python tools/nwn_sweep.py [NPROF=768] [NWN ...]
Wall time of rt.run() on a batch (host arrays in and out: PCIe-inclusive, the same for both kernels), best of 5, per kernel
option wn / ms / auto - a check of api.hip's choice (its cost model is calibrated on 50 channels)."""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from monortm_amd import api, synth, tape3  # noqa: E402


def main():
    nprof = int(sys.argv[1]) if len(sys.argv) > 1 else 768
    nwns = [int(x) for x in sys.argv[2:]] or [10, 20, 32, 50, 64]
    rec = synth.synthetic_lines(500)
    with tempfile.TemporaryDirectory() as wd:
        t3 = os.path.join(wd, "TAPE3")
        tape3.write_tape3(t3, rec)
        for nwn in nwns:
            wn = synth.c2_channels(nwn)
            profs = [synth.perturbed_profile(i, wn, nlay=64) for i in range(nprof)]
            row = []
            for k in ("wn", "ms", "auto"):
                rt = api.MonoRTM(t3, wn[0], wn[-1])
                rt.set_option("lines_kernel", k)
                rt.run(profs)
                best = 1e9
                for _ in range(5):
                    t0 = time.perf_counter()
                    rt.run(profs)
                    best = min(best, time.perf_counter() - t0)
                rt.close()
                row.append(best * 1e3)
            print(f"nwn {nwn:3d} x {nprof} profiles: wn {row[0]:8.3f} ms  ms {row[1]:8.3f} ms  auto {row[2]:8.3f} ms", flush=True)


if __name__ == "__main__":
    main()
