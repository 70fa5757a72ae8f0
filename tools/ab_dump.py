#!/usr/bin/env python3
"""Dump the outputs of one bench workload (tools/ab_dump.py WORKLOAD OUT.npz) with the library named by MONORTM_HIP_LIB:
two runs with two builds, then tools/ab_dump.py --compare A.npz B.npz prints the largest relative differences."""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if sys.argv[1] == "--compare":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    for k in a.files:
        x, y = a[k], b[k]
        scale = np.maximum(np.abs(y), 1e-12 * np.abs(y).max())
        print(f"{k:10s} max rel diff {np.max(np.abs(x - y) / scale):.3e}")
    sys.exit(0)

import bench  # noqa: E402
from monortm_amd import api, tape3  # noqa: E402

rec, profs, desc, _rk = bench.build_workload(sys.argv[1], 0, 8)
d = tempfile.mkdtemp()
t3 = os.path.join(d, "TAPE3")
tape3.write_tape3(t3, rec)
rt = api.MonoRTM(t3, profs[0].wn[0], profs[0].wn[-1])
b = api.DeviceBatch(rt, profs)
b.step()
b.check()
np.savez(sys.argv[2], o=b.O.cpu().numpy(), obm=b.OBM.cpu().numpy(), rad=b.RAD.cpu().numpy(), tb=b.TB.cpu().numpy())
