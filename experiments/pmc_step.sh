# Counters of the gradient-apply step kernels, before (WM_STEP_TILE=0: round 1's wave-per-run step_short_kernel) and after
# (step_tile_kernel), same call: bench.py --op grad_apply (10 M uniform gradient rows, SGD). One counter group per rocprofv3
# pass (kernel trace only). Output: gpurun_out/step_kernel_counters.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/step_kernel_counters.txt
: > $OUT
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum" "TCC_TAG_STALL_sum TCC_REQ_sum" "MemUnitStalled MemUnitBusy" "WriteUnitStalled" "OccupancyPercent" "MeanOccupancyPerCU"; do
  for tile in 0 1; do
    d=/tmp/pmcstep_${tile}_$(echo $grp | tr ' ' '_' | cut -c1-40)
    rm -rf $d
    WM_STEP_TILE=$tile timeout 400 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $d -- python $R/bench.py --op grad_apply --no-cpu-baseline --steps 3 --warmup 2 --stability-steps 0 > /dev/null 2>&1
    f=$(find $d -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python - "$f" "$grp" $tile >> $OUT <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "step_short_kernel" in k or "step_tile_kernel" in k:
        a = agg[k.replace("void wm::(anonymous namespace)::", "").split("<")[0]][r["Counter_Name"]]
        a[0] += 1; a[1] += float(r["Counter_Value"])
for k, cs in agg.items():
    print("%-48s %s" % (k[:48], "  ".join("%s=%.4g" % (c, v / n) for c, (n, v) in sorted(cs.items()))))
if not agg:
    print("(no step kernel rows for group '%s', WM_STEP_TILE=%s — counter not available?)" % (sys.argv[2], sys.argv[3]))
PY
  done
done
cat $OUT
