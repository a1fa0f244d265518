cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_exchange_optim_gpu.py -x -q -m gpu 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
prof() {
  rm -rf /tmp/prof_x; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_x -- python $R/bench.py --op grad_apply --memory-type distributed --no-cpu-baseline --steps 5 --warmup 2 --dist zipf > /dev/null 2>&1
  f=$(find /tmp/prof_x -name "*kernel_stats.csv" | head -1)
  python - $f <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:3]:
    if "step_" in r["Name"]: print(r["Name"][:80], "calls", r["Calls"], "avg_ms", round(float(r["AverageNs"]) / 1e6, 3))
PY
}
prof
