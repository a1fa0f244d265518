#!/bin/bash
# round 5: append_unique's table insert with a look before the compare-and-swap (WM_AU_DIRECT_CAS=0, round 4) against the
# compare-and-swap alone (=1); default = by the number of keys. Fresh processes alternating, then kernel stats of both.
cd "$(dirname "$0")/.."
for v in "WM_AU_DIRECT_CAS=0" "WM_AU_DIRECT_CAS=1" "X=default" "WM_AU_DIRECT_CAS=0" "WM_AU_DIRECT_CAS=1" "X=default" "WM_AU_DIRECT_CAS=0" "WM_AU_DIRECT_CAS=1" "X=default"; do
  r=$(env $v timeout 300 python bench.py --op sample_gather --steps 200 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['ms_per_step'], r.get('stability',{}).get('median_ms'))")
  echo "$v  ms_per_step, median: $r"
done
for v in "WM_AU_DIRECT_CAS=0" "WM_AU_DIRECT_CAS=1"; do
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_cas && env $v timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cas -- python $OLDPWD/bench.py --op sample_gather --steps 50 --stability-steps 0 > /dev/null 2>&1 )
  echo "== $v: kernel stats (name, calls, total ns, average ns)"
  python3 - $(find /tmp/prof_cas -name "*kernel_stats.csv" | head -1) <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    print("%-70s %6s %12s %10.1f" % (r["Name"][:70], r["Calls"], r["TotalDurationNs"], float(r["AverageNs"])))
PY
done
for v in "WM_AU_DIRECT_CAS=1" "WM_AU_DIRECT_CAS=0" "X=default"; do
  r=$(env $v timeout 300 python bench.py --op sample_gather --seeds 65536 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['ms_per_step'], r.get('stability',{}).get('median_ms'))")
  echo "$v  65536 seeds: ms_per_step, median: $r"
done
for v in "WM_AU_DIRECT_CAS=1" "WM_AU_DIRECT_CAS=0" "X=default"; do
  r=$(env $v timeout 300 python bench.py --op sample_gather --seeds 4096 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['ms_per_step'], r.get('stability',{}).get('median_ms'))")
  echo "$v  4096 seeds: ms_per_step, median: $r"
done
