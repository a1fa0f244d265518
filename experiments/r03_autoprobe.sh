#!/bin/bash
# automatic placement probe of wholememory_malloc (WM_MALLOC_PROBE unset) against the probe switched off (=1): scatter and SGD
# gradient apply of the C2-shaped table in back-to-back processes (the placement of a plain allocation varies with what the
# previous process still holds)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03_autoprobe.txt
: > $O
line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("   ms_per_step %.4f  frac %s  write-side probe %s" % (d["ms_per_step"], (d.get("roofline") or {}).get("frac"), (d.get("table_probe") or {}).get("read_write_back_ms_per_GiB")))
except Exception as e:
    print("   no line:", e)
PY
}
for rnd in 1 2 3; do
  for op in scatter grad_apply; do
    for mode in off auto; do
      if [ $mode = off ]; then export WM_MALLOC_PROBE=1; else unset WM_MALLOC_PROBE; fi
      echo "round $rnd  $op  probe $mode" >> $O
      WM_MALLOC_PROBE_VERBOSE=1 timeout 600 python bench.py --op $op --no-cpu-baseline --steps 50 --stability-steps 0 > /tmp/l.json 2> /tmp/l.err
      grep "malloc probe" /tmp/l.err | sed 's/^/   /' >> $O
      line /tmp/l.json >> $O
    done
  done
done
cat $O
