#!/usr/bin/env python3
"""Per-stage cycle stamps of lines_kernel / finish_mw_kernel and per-class statistics of the evaluate stage, read from the
instrumented builds of tools/debug_builds.sh (GPU box):

    MONORTM_HIP_LIB=$PWD/build_dbg/libmonortm_hip_ltiming.so python tools/stage_timing.py lines c4shard 128
    MONORTM_HIP_LIB=$PWD/build_dbg/libmonortm_hip_classes.so python tools/stage_timing.py classes c3 1
    MONORTM_HIP_LIB=$PWD/build_dbg/libmonortm_hip_timing.so  python tools/stage_timing.py finish c4shard 128

The instrumented kernels park their s_memtime differences in the cloud optical-depth output (which is therefore wrong in
these builds): elapsed cycles of ONE wave per workgroup, i.e. including the time it waits for its three SIMD neighbours -
shares of a stage, not instruction counts (those: tools/ablation_pmc.sh)."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    import torch

    from monortm_amd import api

    what, name, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
    res = bench.Resident(name, 0, 0, n)
    b = res.batch
    np.set_printoptions(linewidth=200, suppress=True)
    if what == "classes":
        lib = api.load_library()
        out = (ctypes.c_ulonglong * 32)()
        b.step()
        torch.cuda.synchronize()
        lib.monortm_dbg_stats(out, 1)
        b.step()
        torch.cuda.synchronize()
        lib.monortm_dbg_stats(out, 0)
        a = np.array(list(out), dtype=np.float64)
        names = ["V general", "Y coupled", "skip(far)", "M2+AL", "M2+TEST", "AL", "TEST"]
        tot = a[:7].sum()
        print(name, "class: share of evaluate wave-cycles | sub-runs | lines | cycles per line | per sub-run | mean length")
        for i, nm in enumerate(names):
            if a[8 + i]:
                print(f"{nm:10s} {a[i] / tot:6.3f} {int(a[8 + i]):10d} {int(a[16 + i]):12d} {a[i] / max(a[16 + i], 1):8.1f} "
                      f"{a[i] / a[8 + i]:8.1f} {a[16 + i] / a[8 + i]:6.1f}")
        return
    for _ in range(3):
        b.step()
    torch.cuda.synchronize()
    o = b.OCLW.cpu().numpy()  # [profile, layer, wn]
    if what == "finish":
        t = o.reshape(-1, o.shape[-1])[:, :6]
        print("finish_mw_kernel, mean cycles per workgroup [set-up, A coarse, B XINT, C wavenumbers, D totals, all]:\n", np.round(t.mean(0)))
    else:
        if o.shape[-1] >= 20:   # when did the workgroups run?  (start / end on the 100 MHz counter every CU shares)
            st, en = o[:, :, 18].ravel(), o[:, :, 19].ravel()
            ok = st > 0
            t0 = st[ok].min()
            st, en = (st[ok] - t0) / 100.0, (en[ok] - t0) / 100.0   # microseconds
            print(f"workgroups {ok.sum()}: kernel span {en.max():.1f} us; starts: median {np.median(st):.1f}, 90 % {np.percentile(st, 90):.1f}, "
                  f"max {st.max():.1f}; durations: median {np.median(en - st):.1f}, 10 % {np.percentile(en - st, 10):.1f}, 90 % {np.percentile(en - st, 90):.1f}")
            edges = np.arange(0.0, en.max() + 10.0, 10.0)
            running = [(np.sum((st < b) & (en > a))) for a, b in zip(edges[:-1], edges[1:])]
            print("workgroups alive per 10 us interval:", running)
            first = st < 5.0
            print(f"first round ({first.sum()} workgroups): ends {np.percentile(en[first], 5):.1f} .. {np.percentile(en[first], 95):.1f} us; "
                  f"later workgroups: durations median {np.median((en - st)[~first]):.1f} us")
        t = o[:, :, 8:18]
        print("lines_kernel, mean per sampled workgroup [prologue, prepare, evaluate, all (cycles) | lines, slice lines, far, AL, M2, V "
              "(wave 0)]:\n", np.round(t.reshape(-1, 10).mean(0)))
        for lay in range(0, t.shape[1], 4):
            print(lay, np.round(t[:, lay].mean(0)).astype(int))


if __name__ == "__main__":
    main()
