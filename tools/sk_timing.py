"""Per-phase cycle stamps of lines_state_kernel (debug build -DSK_TIMING, tools/debug_builds.sh): one step of a bench workload;
the kernel prints one line per wave of two of its workgroups.
    MONORTM_LINES_KERNEL=state MONORTM_HIP_LIB=$PWD/build_dbg/libmonortm_hip_sktiming.so python tools/sk_timing.py c4full"""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench  # noqa: E402

res = bench.Resident(sys.argv[1], 0, 0, 128)
res.batch.step()
torch.cuda.synchronize()
