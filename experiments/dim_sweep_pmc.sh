# Row-shape sweep with HBM traffic: dim_sweep.py once for the timings (CSV), then under rocprofv3 --pmc FETCH_SIZE and
# --pmc WRITE_SIZE (separate passes, kernel trace only) for the bytes each gather / scatter launch really moved.
# Output: gpurun_out/dim_sweep.csv (timings) and gpurun_out/dim_sweep_traffic.csv (per shape / op: KiB counters, corrected bytes,
# traffic over algorithmic). FETCH_SIZE counts half of the bytes of 16 B/lane reads on gfx950 (MI355X_MICROARCH.md, HBM section;
# calibrated in profiles/pmc_traffic.json), so it is doubled.
R=$GRAFT_REPO_ROOT
DIMS="${DIMS:-32 64 100 128 129 200 256 300 512 602 1024}"
cd $R && python experiments/dim_sweep.py --csv=$R/gpurun_out/dim_sweep.csv $DIMS > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/ds_$c
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/ds_$c -- python $R/experiments/dim_sweep.py $DIMS > /dev/null 2>&1
done
python - "$R/gpurun_out/dim_sweep.csv" "$R/gpurun_out/dim_sweep_traffic.csv" <<'PY'
import csv, glob, sys
def groups(counter):
    f = glob.glob("/tmp/ds_%s/**/*counter_collection.csv" % counter, recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "rows_" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    vals = [float(r["Counter_Value"]) for r in rows]
    return [sum(vals[i:i + 13]) / 13 for i in range(0, len(vals) - 12, 13)]     # 3 warm-up + 10 timed launches per (shape, op)
fetch, write = groups("FETCH_SIZE"), groups("WRITE_SIZE")
timing = list(csv.DictReader(open(sys.argv[1])))
assert len(fetch) == len(write) == len(timing), (len(fetch), len(write), len(timing))
w = csv.writer(open(sys.argv[2], "w"))
w.writerow(["op", "dtype", "dim", "row_bytes", "n_ids", "ms", "algorithmic_GB", "frac_of_8TBps_algorithmic", "FETCH_SIZE_KiB", "WRITE_SIZE_KiB",
            "read_GB(x2)", "write_GB", "traffic_over_algorithmic", "frac_of_8TBps_traffic"])
for t, f, wr in zip(timing, fetch, write):
    rd, wb = f * 1024 * 2 / 1e9, wr * 1024 / 1e9
    algo, ms = float(t["algorithmic_GB"]), float(t["ms_min"])
    w.writerow([t["op"], t["dtype"], t["dim"], t["row_bytes"], t["n_ids"], t["ms_min"], "%.3f" % algo, t["frac_of_8TBps"], "%.0f" % f, "%.0f" % wr,
                "%.3f" % rd, "%.3f" % wb, "%.3f" % ((rd + wb) / algo), "%.4f" % ((rd + wb) / (ms * 1e-3) / 8000.0)])
PY
cat $R/gpurun_out/dim_sweep_traffic.csv
