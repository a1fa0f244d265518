#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r04_full_tests.txt 2>&1; echo "tests rc $?" >> gpurun_out/r04_full_tests.txt
tail -4 gpurun_out/r04_full_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for op in gather scatter grad_apply; do
timeout 600 python bench.py --op $op --no-cpu-baseline --steps 100 --stability-steps 0 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read()); r = d.get('roofline') or {}
print('$op: ms_per_step %.4f kernel_ms %s frac %s kernel %s' % (d['ms_per_step'], r.get('kernel_ms'), r.get('frac'), (r.get('kernel') or '')[:80]))"
done
