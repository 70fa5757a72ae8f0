cd $GRAFT_REPO_ROOT
for K in wn state; do
  for NS in "" ; do
  echo "== $K"
  MONORTM_LINES_KERNEL=$K python bench.py --no-extra --no-pmc --no-cpu-baseline --steps 200 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d.get('kernel_ms_per_step'))"
  done
done
for NS in 1 2 4; do echo "== state nslice $NS"; MONORTM_NSLICE=$NS MONORTM_LINES_KERNEL=state python bench.py --no-extra --no-pmc --no-cpu-baseline --steps 200 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d.get('kernel_ms_per_step'))"; done
