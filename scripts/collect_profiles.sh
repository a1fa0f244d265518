#!/bin/bash
# Round profiles for profiles/: run on the GPU box through `gpurun -- 'bash scripts/collect_profiles.sh r02'`.
#  1 the contract bench line (default flags) + rocprofv3 --kernel-trace --stats of the same command
#  2 HBM traffic of the gather kernel: separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (kernel trace only), plus the same
#    two passes over a streaming-copy kernel of known size for calibration -> profiles/pmc_traffic.json (scripts/collect_traffic.py)
#  3 side ops: scatter, gradient apply (uniform / zipf, SGD and LazyAdam): bench lines + kernel stats; PMC of the step kernel
#  4 C5 (sample_gather) kernel stats, the C++ bench tool
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd $R
python bench.py > $OUT/${TAG}_n1_bench.json 2> $OUT/${TAG}_n1_bench.err
cut -c1-400 $OUT/${TAG}_n1_bench.json
stats() {  # stats <name> <bench args...>: bench line + kernel stats of the same command
  local name=$1; shift
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_$name && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -- python $R/bench.py "$@" > $OUT/${TAG}_${name}_under_rocprof.json 2>/dev/null )
  cp $(find /tmp/prof_$name -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_${name}_kernel_stats.csv
  head -4 $OUT/${TAG}_${name}_kernel_stats.csv | cut -c1-160
}
stats n1_bench --no-cpu-baseline
# the same process, per launch: the placement probes of bench.py run on slower candidate pairs too, so the stats average over
# ALL launches sits above the timed region's; the tail of the trace (timed region + stability leg) is what `kernel_ms` measures
WM_TAG=$TAG WM_KERNEL_MS_OUT=$OUT/kernel_ms.json WM_BENCH_LINE=$OUT/${TAG}_n1_bench_under_rocprof.json python - $(find /tmp/prof_n1_bench -name "*kernel_trace.csv" | head -1) > $OUT/${TAG}_n1_bench_kernel_trace_tail.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "rows_batch_kernel" in r["Kernel_Name"] or "rows_copy16_fast_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
tail = dur[-400:]
print("kernel: %s" % rows[-1]["Kernel_Name"])
import json, os
step_ms = None
try:   # the step time of THIS collection's own bench line: bench.py cites the kernel time only for runs at the same speed
    step_ms = json.loads(open(os.environ["WM_BENCH_LINE"]).read().strip().splitlines()[-1])["ms_per_step"]
except Exception:
    pass
json.dump({"file": os.environ.get("WM_TAG", "r05") + "_n1_bench_kernel_stats.csv", "commit": os.environ.get("WM_COMMIT", "worktree"),
           "collection_ms_per_step": step_ms,
           "kernel": rows[-1]["Kernel_Name"], "launches": len(tail), "average_ms": round(sum(tail) / len(tail), 4),
           "min_ms": round(min(tail), 4), "max_ms": round(max(tail), 4),
           "what": "rocprofv3 --kernel-trace of `python bench.py --no-cpu-baseline`: the last launches of the process "
                   "(timed region + stability leg)"}, open(os.environ["WM_KERNEL_MS_OUT"], "w"), indent=1)
print("all %d launches of the process (placement probes included): average %.4f ms" % (len(dur), sum(dur) / len(dur)))
print("last %d launches (timed region + stability leg, on the picked pair): average %.4f ms, min %.4f, max %.4f" % (
    len(tail), sum(tail) / len(tail), min(tail), max(tail)))
PY
cat $OUT/${TAG}_n1_bench_kernel_trace_tail.txt
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pmc_$c && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- python $R/bench.py --no-cpu-baseline --steps 10 --warmup 2 --stability-steps 0 > /dev/null 2>&1 )
  cp $(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1) $OUT/${TAG}_bench_${c}_counter_collection.csv
  if [ -x $R/experiments/gather_variants ]; then
    ( cd /tmp && rm -rf /tmp/cal_$c && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/cal_$c -- $R/experiments/gather_variants 100000000 10000000 3 pmc > /dev/null 2>&1 )
    f=$(find /tmp/cal_$c -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${TAG}_calib_${c}_counter_collection.csv
  fi
done
python scripts/collect_traffic.py $OUT $TAG && cat $OUT/pmc_traffic.json | head -30
python bench.py --op scatter --no-cpu-baseline > $OUT/${TAG}_scatter_bench.json 2>/dev/null
stats scatter --op scatter --no-cpu-baseline --steps 50 --stability-steps 0
for d in uniform zipf; do for o in sgd adam; do
  python bench.py --op grad_apply --dist $d --optimizer $o --no-cpu-baseline > $OUT/${TAG}_grad_apply_${o}_${d}_bench.json 2>/dev/null
  cut -c1-200 $OUT/${TAG}_grad_apply_${o}_${d}_bench.json | head -1
done; done
stats grad_apply --op grad_apply --no-cpu-baseline --steps 30 --stability-steps 0
stats grad_apply_zipf --op grad_apply --dist zipf --no-cpu-baseline --steps 30 --stability-steps 0
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pmcg_$c && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmcg_$c -- python $R/bench.py --op grad_apply --no-cpu-baseline --steps 5 --warmup 2 --stability-steps 0 > /dev/null 2>&1 )
  python - $(find /tmp/pmcg_$c -name "*counter_collection.csv" | head -1) $c >> $OUT/${TAG}_grad_apply_pmc_per_kernel.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:90]
    agg[k][0] += 1
    agg[k][1] += float(r["Counter_Value"])
print("==", sys.argv[2], "(KiB per launch; FETCH_SIZE counts half of the bytes of 16 B/lane reads on gfx950)")
for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:8]:
    print("%-92s launches %4d  KiB/launch %14.1f" % (k, n, v / n))
PY
done
cat $OUT/${TAG}_grad_apply_pmc_per_kernel.txt | head -24
python bench.py --op sample_gather --steps 50 --stability-steps 50 > $OUT/${TAG}_sample_gather_bench.json 2>/dev/null
stats sample_gather --op sample_gather --steps 50 --stability-steps 0
if [ -x tools/gather_scatter_bench ]; then
  for f in gather scatter; do
    echo "\$ tools/gather_scatter_bench -t chunked -l device -e 51200000000 -g 5120000000 -d 128 -c 20 -p 6 -f $f" >> $OUT/${TAG}_cpp_bench.txt
    tools/gather_scatter_bench -t chunked -l device -e 51200000000 -g 5120000000 -d 128 -c 20 -p 6 -f $f >> $OUT/${TAG}_cpp_bench.txt 2>&1
  done
  tail -8 $OUT/${TAG}_cpp_bench.txt
fi
ls $OUT
