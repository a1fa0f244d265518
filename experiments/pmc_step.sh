cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -oE "Name:\s*[A-Za-z0-9_]+|^\s*[A-Z][A-Z0-9_]+\s" | tr -s ' ' | sort -u | grep -E "TCC_(EA|HIT|MISS|REQ|TAG|BUSY|STALL|WRITE|READ)|TCP_(PENDING|TCC|TOTAL|TA)|SQ_(WAVES|BUSY|WAIT|INSTS_VMEM|ACTIVE_INST_VMEM|INST_CYCLES_VMEM|LEVEL)|GRBM_GUI|MemUnit|L2Cache|WriteUnitStalled" | head -80
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum" "MemUnitStalled WriteUnitStalled MemUnitBusy"; do
  d=$R/gpurun_out/pmcstep_$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout 400 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $d -- python $R/bench.py --op grad_apply --memory-type distributed --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2>&1
  f=$(find $d -name "*counter_collection.csv" | head -1)
  echo "== $grp"
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "step_short" in k or "rows_copy16" in k:
        a = agg[k[:60]][r["Counter_Name"]]
        a[0] += 1; a[1] += float(r["Counter_Value"])
for k, cs in agg.items():
    print(k, {c: round(v / n, 1) for c, (n, v) in cs.items()})
PY
done
