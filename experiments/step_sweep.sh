cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cp $R/wholegraph_amd/libwholegraph.so /tmp/lib_orig.so
prof() {
  rm -rf /tmp/prof_x; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_x -- python $R/bench.py --op grad_apply --memory-type distributed --no-cpu-baseline --steps 5 --warmup 2 --dist zipf > /dev/null 2>&1
  f=$(find /tmp/prof_x -name "*kernel_stats.csv" | head -1)
  python - $f <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:3]:
    if "step_long" in r["Name"]: print(r["Name"][:80], "calls", r["Calls"], "avg_ms", round(float(r["AverageNs"]) / 1e6, 3))
PY
}
echo "== S32 (default)"; prof
for v in S16 S64; do cp $R/experiments/variants/libwholegraph_$v.so $R/wholegraph_amd/libwholegraph.so; echo "== $v"; prof; done
cp /tmp/lib_orig.so $R/wholegraph_amd/libwholegraph.so
