cd /tmp && export TMPDIR=/tmp
for only in 0 11 12 13; do
  rm -rf /tmp/hs$only
  SPLIT_HOT=1 SPLIT_FIX_ONLY=$only timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/hs$only -- $GRAFT_REPO_ROOT/experiments/split_sort_test big > /dev/null 2>&1
  python3 - $(find /tmp/hs$only -name "*kernel_trace.csv" | head -1) $only <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the LAST launch of each kernel = the Zipf case (single launch when the result mismatches, else the last of the timing loop)
last = {}
for r in rows:
    last[r["Kernel_Name"][:60]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("only=%s:" % sys.argv[2], "  ".join("%s %.1f" % (k.replace("void wm::split::","").replace("wm::split::","")[:28], v) for k, v in last.items() if "split" in k or "hot_select" in k))
PY
done
