#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03/scatter
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for i in 1 2 3; do timeout 300 $R/experiments/scatter_tables 4 6 > $OUT/plain_$i.txt 2>&1; cat $OUT/plain_$i.txt; echo; done
i=0
for set in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_WRITEBACK_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_TCC_WRITE_REQ_LATENCY_sum" \
           "TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_NORMAL_WRITEBACK_sum TCC_NORMAL_EVICT_sum"; do
  i=$((i+1)); rm -rf /tmp/st_$i
  timeout 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/st_$i -- $R/experiments/scatter_tables 4 3 > $OUT/pmc_${i}_stdout.txt 2>&1
  f=$(find /tmp/st_$i -name "*counter_collection.csv" | head -1)
  echo "== pass $i: $set"; grep "round\|table" $OUT/pmc_${i}_stdout.txt | head -4
  [ -n "$f" ] && python3 - $f <<'PY' | tee $OUT/pmc_${i}_summary.txt
import csv, re, sys, collections
v = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"(rows8|fill_rows_random)<(\d+)(?:, (true|false), (\d+))?>", r["Kernel_Name"])
    if not m: continue
    key = "%s table %s %s policy %s" % (m.group(1), m.group(2), {"true": "scatter", "false": "gather", None: ""}[m.group(3)], m.group(4) or "-")
    v[key][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[key].add(r["Dispatch_Id"])
for key in sorted(v):
    if "policy 0" in key or "fill" in key:
        print("%-46s" % key, "  ".join("%s %.4g" % (c.replace("_sum", ""), x / len(cnt[key])) for c, x in sorted(v[key].items())))
PY
done
