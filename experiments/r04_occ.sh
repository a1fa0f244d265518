#!/bin/bash
# occupancy sweep of the single-batch kernel through unused dynamic LDS (WM_ROWS_LDS bytes per one-wave workgroup):
# waves per CU = floor(160 KiB / bytes), capped at 24 by the kernel's waves_per_eu(1, 6)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04_occ_${1:-product}.txt
: > $O
for rep in 1 2; do
for lds in 0 40960 27296 20480 16384 13648 10240 8192 6816; do
  for op in gather scatter; do
    WHOLEGRAPH_AMD_VARIANT=${1} WM_ROWS_LDS=$lds timeout 600 python bench.py --op $op --no-cpu-baseline --steps 100 --stability-steps 0 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read())
r = d.get('roofline') or {}
lds = $lds
print('%-8s lds %6d (%s waves/CU)  ms_per_step %.4f  kernel_ms %s frac %s' % ('$op', lds, (163840 // lds) if lds else 'cap', d['ms_per_step'], r.get('kernel_ms'), r.get('frac')))
" >> $O
  done
done
done
cat $O
