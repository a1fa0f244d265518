#!/bin/bash
# the gradient apply's --pmc passes of scripts/collect_profiles.sh alone (counter collection lets one kernel run at a time)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/profiles_r05; mkdir -p $OUT; cd $R
: > $OUT/r05_grad_apply_pmc_per_kernel.txt
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pmcg_$c && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmcg_$c -- python $R/bench.py --op grad_apply --no-cpu-baseline --steps 5 --warmup 2 --stability-steps 0 > /dev/null 2>&1 < /dev/null; echo "rc $?" )
  f=$(find /tmp/pmcg_$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] || { echo "no counter file for $c"; continue; }
  python - $f $c >> $OUT/r05_grad_apply_pmc_per_kernel.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:90]
    agg[k][0] += 1
    agg[k][1] += float(r["Counter_Value"])
print("==", sys.argv[2], "(KiB per launch; FETCH_SIZE counts half of the bytes of 16 B/lane reads on gfx950)")
for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:8]:
    print("%-92s launches %4d  KiB/launch %14.1f" % (k, n, v / n))
PY
done
cat $OUT/r05_grad_apply_pmc_per_kernel.txt
