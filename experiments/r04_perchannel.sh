#!/bin/bash
# per-TCC-channel write counters of the scatter on four 51.2 GB tables alive in one process: does a slow table load its channels unevenly?
# (repeated until a process has a slow table among its four, at most 5 times)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for try in 1 2 3 4 5; do
  rm -rf /tmp/pc1
  BLOCKERS_GB=$((try * 7)) PROBE=0 ROUNDS=1 REPS=3 timeout 900 rocprofv3 --pmc TCC_EA0_WRREQ TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_EA0_WRREQ_LEVEL TCC_TAG_STALL --kernel-trace --output-format json -d /tmp/pc1 -- python $R/experiments/tables_in_one_process.py 4 > /tmp/pc1.log 2>&1
  grep "round 0" /tmp/pc1.log | cut -c1-110
  spread=$(grep "round 0" /tmp/pc1.log | python3 -c "
import sys,re
v=[float(re.search(r'scatter ([0-9.]+) ms', l).group(1)) for l in sys.stdin]
print(1 if v and max(v)/min(v) > 1.08 else 0)")
  if [ "$spread" = 1 ]; then
    python3 $R/experiments/perchannel_reduce.py $(find /tmp/pc1 -name "*results.json" | head -1)
    exit 0
  fi
done
echo "no slow table in 5 processes"
