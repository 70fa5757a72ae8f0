# tools/scratchmap.py - which source lines the scratch (spill) instructions of lines_ms_kernel<false> come from: compile
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 -S -gline-tables-only --cuda-device-only -mllvm -disable-machine-licm monortm_amd/csrc/lines_ms_kernel.hip -o /tmp/lines_ms_g.s
# then run this script (LABNOTES 5b).
import re,collections,sys
fn=None; cur=None; cnt=collections.Counter()
for ln in open('/tmp/lines_ms_g.s'):
    m=re.match(r'^(_Z\w+):',ln)
    if m: fn=m.group(1)
    m=re.match(r'\s*\.loc\s+(\d+)\s+(\d+)\s+\d+.*?; (\S+)',ln)
    if m:
        at=re.findall(r'@\[ (\S+?):(\d+):\d+',ln)
        src=m.group(3).split('/')[-1]
        cur=(src.rsplit(':',1)[0] if ':' in src else src, at[-1][0].split('/')[-1]+':'+at[-1][1] if at else '')
    t=ln.strip()
    if t.startswith('scratch_') and fn and 'lines_ms_kernelILb0' in fn: cnt[cur+(t.split()[0].replace('_dwordx2','').replace('_dwordx4','').replace('_dword',''),)]+=1
for k,v in sorted(cnt.items(), key=lambda x:-x[1])[:40]: print(v,k)
