cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cp $R/wholegraph_amd/libwholegraph.so /tmp/lib_orig.so
prof() {
  rm -rf /tmp/prof_x; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_x -- python $R/bench.py --op grad_apply --memory-type distributed --no-cpu-baseline --steps 5 --warmup 2 $1 > /dev/null 2>&1
  f=$(find /tmp/prof_x -name "*kernel_stats.csv" | head -1)
  python - $f <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "step_short" in r["Name"]:
        print(r["Name"][:70], "avg_ms", round(float(r["AverageNs"]) / 1e6, 3))
PY
}
for v in K4 K8; do cp $R/experiments/variants/libwholegraph_$v.so $R/wholegraph_amd/libwholegraph.so
echo "== $v vec4 path"; prof; echo "== $v vec2 path"; WM_STEP_NO_VEC4=1 prof; done
cp /tmp/lib_orig.so $R/wholegraph_amd/libwholegraph.so
