#!/bin/bash
# per-kernel times of the onesweep harness
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r03
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/osw
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/osw -- $R/experiments/onesweep_ab ${1:-5} ${2:--1} ${3:-0} > /tmp/osw_out.txt 2>&1
python - $(find /tmp/osw -name "*kernel_stats.csv" | head -1) > $R/gpurun_out/r03/onesweep_kernel_stats.txt <<'PY'
import csv, sys, re
for r in csv.DictReader(open(sys.argv[1])):
    n = re.sub(r"rocprim::ROCPRIM_\d+_NS::detail::", "rp::", r["Name"])
    n = re.sub(r"^void ", "", n)
    print("%6s calls  avg %9.1f us  min %9.1f  max %9.1f  %s" % (r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, n[:150]))
PY
cat $R/gpurun_out/r03/onesweep_kernel_stats.txt
