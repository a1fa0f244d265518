#!/bin/bash
# what bounds sample_small_kernel on the second hop of a C5 step? instruction counts and wait cycles per launch (rocprofv3 --pmc,
# one pass per group; kernel trace only)
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES"; do
  rm -rf /tmp/pmc_s
  timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pmc_s -- python $GRAFT_REPO_ROOT/bench.py --op sample_gather --steps 6 --warmup 2 --stability-steps 0 > /dev/null 2>&1
  f=$(find /tmp/pmc_s -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    for name in ("sample_small_kernel", "au_insert_kernel"):
        if name in r["Kernel_Name"]:
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    for c, v in cs.items():
        v.sort()
        print("%-42s %-22s launches %3d  min %14.0f  max %14.0f" % (k, c, len(v), v[0], v[-1]))
PY
done
