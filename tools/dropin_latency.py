#!/usr/bin/env python3
"""Where the time of the one-profile-per-call pattern goes (GPU box): the three host-buffer C ABI calls timed through
ctypes with preallocated arrays, next to the Fortran harness (same calls through the ISO_C_BINDING shim).

    python tools/dropin_latency.py [nprof]
"""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from monortm_amd import _build, api, caseio, synth, tape3  # noqa: E402

nprof = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rec = synth.synthetic_lines(500)
wn = synth.c2_channels(50)
profs = [synth.perturbed_profile(i, wn, nlay=64) for i in range(nprof)]
tmp = tempfile.mkdtemp()
t3 = os.path.join(tmp, "TAPE3")
tape3.write_tape3(t3, rec)
rt = api.MonoRTM(t3, wn[0], wn[-1])
for rep in range(2):
    tm = tr1 = tr2 = 0.0
    for p in profs:
        t0 = time.perf_counter()
        O, OBM, OC, OCLW = rt.modm([p])
        t1 = time.perf_counter()
        a = rt.rtm([p], O)
        t2 = time.perf_counter()
        b = rt.rtm([p], O)
        t3_ = time.perf_counter()
        tm += t1 - t0
        tr1 += t2 - t1
        tr2 += t3_ - t2
    print(f"pass {rep}: python host calls per profile: modm {tm / nprof * 1e6:.1f} us, rtm(+tmr) {tr1 / nprof * 1e6:.1f} us, "
          f"rtm {tr2 / nprof * 1e6:.1f} us (includes numpy packing); O reused {rt.lib.monortm_hip_counter(rt.ctx, 0)}")
rt.close()
exe = os.path.join(_build.LIBDIR, "harness_hip_dbl")
cp, op = os.path.join(tmp, "case.bin"), os.path.join(tmp, "out.bin")
caseio.write_case(cp, profs)
for rep in range(2):
    r = subprocess.run([exe, cp, t3, op, "3"], cwd=tmp, capture_output=True, text=True)
    print(r.stdout.strip().splitlines()[-1], "->", [float(x.split()[1]) / (3 * nprof) * 1e3 for x in r.stdout.splitlines()
                                                       if x.startswith("HARNESS_SECONDS")], "ms per profile")
