#!/bin/bash
# what the random 512 B READ stream of the gather costs, in counters: the contract gather (uniform ids) against the same kernel on
# sequential ids, one rocprofv3 --pmc pass per counter group and id distribution (kernel trace only), mean per launch of rows_batch_kernel
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_random_read_pmc.txt
: > $O
pass() {
  local name=$1; shift
  for d in uniform sequential; do
    rm -rf /tmp/rr_$name
    timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/rr_$name -- python $R/bench.py --dist $d --no-cpu-baseline --no-check --steps 10 --warmup 2 --stability-steps 0 > /dev/null 2>&1
    f=$(find /tmp/rr_$name -name "*counter_collection.csv" | head -1)
    python3 - "$f" $d >> $O <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
if sys.argv[1]:
    for r in csv.DictReader(open(sys.argv[1])):
        if "rows_batch_kernel" in r["Kernel_Name"]:
            a = agg[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k in sorted(agg):
    n, v = agg[k]
    print("%-12s %-44s launches %3d  mean per launch %18.1f" % (sys.argv[2], k, n, v / n))
PY
  done
}
pass utcl1 TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum
pass tcplat TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum
pass tccrd TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum
pass tccwr TCC_LATENCY_FIFO_FULL_sum TCC_IB_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum
pass grbm GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE
pass tcpreq TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_CACHE_ACCESSES_sum
cat $O
