#!/bin/bash
# A/B of the two line-sum kernels on a workload of bench.py (GPU box, repo root): tools/ab_lines_kernel.sh [workload ...]
cd "$GRAFT_REPO_ROOT" || exit 1
for W in ${@:-c4shard c4full}; do
  for K in wn state; do
    echo "== $W $K"
    MONORTM_LINES_KERNEL=$K python bench.py --workload $W --no-extra --no-pmc --no-cpu-baseline --steps 50 --min-seconds 0.5 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%.4g evals/s' % d['value'], 'ms/step %.4f' % d['ms_per_step'], d.get('kernel_ms_per_step'))"
  done
done
