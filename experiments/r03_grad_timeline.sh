#!/bin/bash
# round 3: per-call timeline of the uniform gradient apply (kernel trace): kernel time, gaps between launches, the sequence
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r03
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/gt
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/gt -- python $R/bench.py --op grad_apply --no-cpu-baseline --steps 20 --warmup 3 --stability-steps 0 > $R/gpurun_out/r03/grad_timeline_bench.json 2>/dev/null
python - $(find /tmp/gt -name "*kernel_trace.csv" | head -1) > $R/gpurun_out/r03/grad_timeline.txt <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = re.sub(r"rocprim::ROCPRIM_\d+_NS::detail::", "rp::", n)
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n[:100]
# calls = sequences ending with step_tile_kernel / step_long kernels; take the last 10 step_tile launches as anchors
idx = [i for i, r in enumerate(rows) if "step_tile_kernel" in r["Kernel_Name"]]
for which in (idx[-3], idx[-2]):
    prev = [j for j in idx if j < which][-1]
    seq = rows[prev + 1: which + 1]
    # the sequence may include trailing kernels of the previous call (step_long*) — keep all, they are part of the period
    t0 = int(rows[prev]["End_Timestamp"])
    print("---- period from the end of one step_tile_kernel to the end of the next: %.1f us" % ((int(rows[which]["End_Timestamp"]) - t0) / 1e3))
    last_end = t0
    ksum = gsum = 0.0
    for r in seq:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print("  gap %7.1f us  run %8.1f us  %s" % ((s - last_end) / 1e3, (e - s) / 1e3, short(r["Kernel_Name"])))
        ksum += (e - s) / 1e3; gsum += max(0, s - last_end) / 1e3
        last_end = max(last_end, e)
    print("  kernels %.1f us, gaps %.1f us, %d launches" % (ksum, gsum, len(seq)))
PY
cat $R/gpurun_out/r03/grad_timeline.txt | head -90
cut -c1-300 $R/gpurun_out/r03/grad_timeline_bench.json
