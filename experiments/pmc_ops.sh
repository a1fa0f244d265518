# PMC traffic of the gradient-apply and scatter kernels (separate passes per counter, kernel-trace only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for op in grad_apply scatter; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${op}_$c -- python $R/bench.py --op $op --memory-type distributed --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2>&1
    f=$(find $R/gpurun_out/pmc_${op}_$c -name "*counter_collection.csv" | head -1)
    echo "== $op $c $f"
    python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:70]
    agg[k][0] += 1
    agg[k][1] += float(r["Counter_Value"])
for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:8]:
    print("%-72s launches %4d  KiB/launch %14.1f" % (k, n, v / n))
PY
  done
done
