#!/bin/bash
# round 5: append_unique's table insert. round4 = look at the slot, then compare-and-swap (WM_AU_MERGE=0 WM_AU_DIRECT_CAS=0);
# direct = compare-and-swap without the look (WM_AU_MERGE=0 WM_AU_DIRECT_CAS=2); merged = the workgroup's keys merged in LDS
# first, then without (WM_AU_DIRECT_CAS=1) / with (=0) the look; default = merged, without the look up to 2 M keys.
# bench.py --op sample_gather in fresh processes, alternating; then kernel stats.
cd "$(dirname "$0")/.."
line() { python3 -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['ms_per_step'], r.get('stability',{}).get('median_ms'))"; }
R4="WM_AU_MERGE=0 WM_AU_DIRECT_CAS=0"; DI="WM_AU_MERGE=0 WM_AU_DIRECT_CAS=2"; MD="WM_AU_DIRECT_CAS=1"; ML="WM_AU_DIRECT_CAS=0"; DF="X=default"
for rep in 1 2 3; do for v in "$R4" "$DI" "$ML" "$DF"; do
  echo "1024 seeds  $v  ms_per_step, median: $(env $v timeout 300 python bench.py --op sample_gather --steps 200 2>/dev/null | line)"
done; done
for seeds in 65536 4096; do for v in "$R4" "$MD" "$ML" "$DF"; do
  echo "$seeds seeds  $v  ms_per_step, median: $(env $v timeout 300 python bench.py --op sample_gather --seeds $seeds 2>/dev/null | line)"
done; done
for v in "$R4" "$DF"; do
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_cas && env $v timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cas -- python $OLDPWD/bench.py --op sample_gather --steps 50 --stability-steps 0 > /dev/null 2>&1 )
  echo "== $v: kernel stats (name, calls, total ns, average ns)"
  python3 - $(find /tmp/prof_cas -name "*kernel_stats.csv" | head -1) <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    if "wm::" in r["Name"]: print("%-70s %6s %12s %10.1f" % (r["Name"][:70], r["Calls"], r["TotalDurationNs"], float(r["AverageNs"])))
PY
done
