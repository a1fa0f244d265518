#!/bin/bash
# step_long4_kernel (ordered fold of the 527 k-row run of the Zipf batch): kernel time (rocprofv3 serialises the two streams: "alone")
# product (128-byte slices: 32 fp32 columns per folding wave) vs variant slice64 (16 columns: one 16-lane pass per VALU instruction?)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "" slice64 "" slice64; do
  rm -rf /tmp/la
  WHOLEGRAPH_AMD_VARIANT=$v timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/la -- python $R/bench.py --op grad_apply --dist zipf --no-cpu-baseline --steps 20 --stability-steps 0 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('${v:-product}: whole call (under rocprof)', d['ms_per_step'], end='  ')"
  python3 - $(find /tmp/la -name "*kernel_stats.csv" | head -1) <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:4]:
    n=r['Name'].replace('wm::(anonymous namespace)::','')
    if 'step_' in n: print('%s %.1f us'%(n[5:22],float(r['AverageNs'])/1e3), end='  ')
print()
PY
  WHOLEGRAPH_AMD_VARIANT=$v timeout 600 python $R/bench.py --op grad_apply --dist zipf --no-cpu-baseline --steps 50 --stability-steps 0 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('      ${v:-product}: whole call, not profiled', d['ms_per_step'])"
done
