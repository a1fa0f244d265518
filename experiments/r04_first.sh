#!/bin/bash
# round 4, first GPU call: the restructured row kernels (straight-line batches) — GPU tests, the contract line, and the
# product (ds_bpermute bases for 512 B rows) against the v_readlane variant, three fresh processes each, gather + scatter
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04_first_tests.txt 2>&1; echo "tests rc $?" >> gpurun_out/r04_first_tests.txt
tail -5 gpurun_out/r04_first_tests.txt
timeout 600 python bench.py > gpurun_out/r04_first_bench.json 2> gpurun_out/r04_first_bench.err; tail -c 1500 gpurun_out/r04_first_bench.json
O=gpurun_out/r04_first_ab.txt
: > $O
for i in 1 2 3; do
  for variant in product readlane; do
    for op in gather scatter; do
      VV=""; [ $variant != product ] && VV=$variant
      WHOLEGRAPH_AMD_VARIANT=$VV timeout 600 python bench.py --op $op --no-cpu-baseline --steps 100 --stability-steps 0 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read())
r = d.get('roofline') or {}
print('%-9s %-8s process %s  ms_per_step %.4f  kernel_ms %s frac %s  kernel %s' % ('$variant', '$op', '$i', d['ms_per_step'], r.get('kernel_ms'), r.get('frac'), (r.get('kernel') or '')[:70]))
" >> $O
    done
  done
done
cat $O
