import sys, tempfile, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import common
from monortm_amd import api
worst={}
with tempfile.TemporaryDirectory() as wd:
    for name in common.golden_names():
        g=common.Golden(name, wd)
        for kern in ("wn","ms"):
            if kern=="ms" and g.profiles[0].nwn>64: continue
            rt=api.MonoRTM(g.tape3, g.profiles[0].wn[0], g.profiles[0].wn[-1]); rt.set_option("lines_kernel",kern)
            for i,(pr,exp) in enumerate(zip(g.profiles,g.expected)):
                got=rt.run([pr])[0]
                e=common.per_molecule_errors(got,exp)
                m=np.nanmax(e) if np.isfinite(e).any() else 0.0
                worst[(name,kern,i)]=(m,int(np.nanargmax(e))+1 if np.isfinite(e).any() else 0)
            rt.close()
for k,v in sorted(worst.items(), key=lambda x:-x[1][0])[:12]: print(k, "%.3e"%v[0], "mol",v[1])
