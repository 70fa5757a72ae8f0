#!/usr/bin/env python3
"""Print the observed relative errors of the HIP path against every golden fixture (outputs of the reference
itself) - the numbers behind the 1e-6 parity claim.  Needs the MI355X.   python tools/parity_report.py"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import Golden, compare, golden_names  # noqa: E402
from monortm_amd import api  # noqa: E402

tmp = tempfile.mkdtemp()
print(f"{'fixture':28s} {'O':>9s} {'O_BY_MOL':>9s} {'OC':>9s} {'RAD':>9s} {'TB':>9s} {'TMR':>9s} {'TRTOT':>9s}")
worst = 0.0
for name in golden_names():
    g = Golden(name, tmp)
    rt = api.MonoRTM(g.tape3, g.profiles[0].wn[0], g.profiles[0].wn[-1])
    agg = {}
    for pr, exp in zip(g.profiles, g.expected):
        e = compare(rt.run([pr])[0], exp, rtol=1e-6, what=name)
        for k, v in e.items():
            agg[k] = max(agg.get(k, 0.0), v)
    rt.close()
    worst = max(worst, max(agg.values()))
    print(f"{name:28s} " + " ".join(f"{agg[k]:9.1e}" for k in ("o", "o_by_mol", "oc", "rad", "tb", "tmr", "trtot")))
print(f"worst relative error over all fixtures and fields: {worst:.2e}   (tolerance 1e-6)")
