#!/bin/bash
# gather of 4000 B rows (flat-stream kernel, rows start on 32-byte multiples) against 4096 B rows (single-batch kernel, rows on
# 4 KiB multiples): where do 8 points go with 2 % more bytes? One rocprofv3 --pmc pass per counter group, mean per gather launch.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_misaligned_pmc.txt
: > $O
pass() {
  local name=$1; shift
  rm -rf /tmp/mp_$name
  timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/mp_$name -- python $R/experiments/dim_sweep.py 1000 1024 > /dev/null 2>&1
  f=$(find /tmp/mp_$name -name "*counter_collection.csv" | head -1)
  python3 - "$f" >> $O <<'PY'
import csv, sys, collections
agg = collections.OrderedDict()
if sys.argv[1]:
    rows = [r for r in csv.DictReader(open(sys.argv[1])) if "rows_" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    for r in rows:
        k = (r["Kernel_Name"].replace("void wm::(anonymous namespace)::", "").split("(")[0][:48], r["Grid_Size"], r["Counter_Name"])
        a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
for (kn, grid, c), (n, v) in agg.items():
    print("%-48s grid %-10s %-40s launches %3d  mean %16.1f" % (kn, grid, c, n, v / n))
PY
}
pass ea TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
pass tcc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum
pass tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
pass stall TCC_EA0_RDREQ_LEVEL_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum
pass busy GRBM_GUI_ACTIVE GRBM_UTCL2_BUSY
pass tlb TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum
cat $O
