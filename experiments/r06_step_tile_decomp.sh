# review item 5 (round 6): step_tile_kernel on four batches (experiments/step_tile_decomp.py), whole-call times in ONE session,
# the kernel's own duration from rocprofv3 --kernel-trace --stats, then the translation / memory-side counters per variant
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_step_tile_decomposition.txt
: > $OUT
echo "== whole call, HIP events, 20 steps (plain allocations; two rounds) ==" >> $OUT
for r in 1 2; do for v in as_is grads_seq table_dense both_seq; do python $R/experiments/step_tile_decomp.py $v 20 2>/dev/null | tail -1 >> $OUT; done; done
echo "== kernel durations (rocprofv3 --kernel-trace --stats), 10 steps ==" >> $OUT
for v in as_is grads_seq table_dense both_seq; do
  d=/tmp/dec_$v; rm -rf $d
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python $R/experiments/step_tile_decomp.py $v 10 > /dev/null 2>&1
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  echo "-- $v" >> $OUT
  python3 - $f >> $OUT <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:7]:
    print("   %-90s calls %4s  avg %9.1f us" % (r['Name'][:90], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
echo "== counters of step_tile_kernel per launch (one group per pass; events-only synchronisation under --pmc) ==" >> $OUT
for grp in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum" "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE" "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_sum" "FETCH_SIZE WRITE_SIZE" "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum"; do
  for v in as_is grads_seq table_dense both_seq; do
    d=/tmp/decpmc_${v}_$(echo $grp | tr ' ' '_' | cut -c1-40); rm -rf $d
    timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $d -- python $R/experiments/step_tile_decomp.py $v 3 > /dev/null 2>&1
    f=$(find $d -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python3 - "$f" "$v" >> $OUT <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if "step_tile_kernel" in r["Kernel_Name"]:
        a = agg[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
print("   %-12s %s" % (sys.argv[2], "  ".join("%s=%.5g" % (c, v / n) for c, (n, v) in sorted(agg.items())) or "(no rows)"))
PY
  done
done
cat $OUT
