#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_contiguous_table.txt
: > $O
for rep in 1 2; do for c in 0 1; do
  WM_MALLOC_CONTIGUOUS=$c timeout 600 python $R/bench.py --no-cpu-baseline --steps 100 --stability-steps 0 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('WM_MALLOC_CONTIGUOUS=$c gather ms_per_step %.4f frac %s  table probe %s' % (d['ms_per_step'], r['frac'], json.dumps({k: v for k, v in (d.get('table_probe') or {}).items() if 'ms_per' in k})))" >> $O
  WM_MALLOC_CONTIGUOUS=$c timeout 600 python $R/bench.py --op scatter --no-cpu-baseline --steps 100 --stability-steps 0 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('WM_MALLOC_CONTIGUOUS=$c scatter ms_per_step %.4f frac %s' % (d['ms_per_step'], r['frac']))" >> $O
done; done
for c in 0 1; do
  rm -rf /tmp/cc_$c
  WM_MALLOC_CONTIGUOUS=$c timeout 600 rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/cc_$c -- python $R/bench.py --no-cpu-baseline --no-check --steps 10 --warmup 2 --stability-steps 0 > /dev/null 2>&1
  f=$(find /tmp/cc_$c -name "*counter_collection.csv" | head -1)
  python3 - "$f" $c >> $O <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if "rows_batch_kernel" in r["Kernel_Name"]:
        a = agg[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k in sorted(agg):
    n, v = agg[k]
    print("WM_MALLOC_CONTIGUOUS=%s %-36s mean per launch %16.1f" % (sys.argv[2], k, v / n))
PY
done
cat $O
