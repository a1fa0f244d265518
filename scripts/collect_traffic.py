#!/usr/bin/env python
"""profiles/pmc_traffic.json from the raw rocprofv3 --pmc CSVs written by scripts/collect_profiles.sh:
HBM bytes per launch of the gather kernel = FETCH_SIZE x fetch_correction + WRITE_SIZE (counter values are KiB), the
correction measured on a streaming-copy kernel of known size in the same session (gfx950's FETCH_SIZE counts half of the
bytes of 16 B/lane reads: /opt/skills/guides/MI355X_MICROARCH.md, HBM section). usage: collect_traffic.py <dir> <tag>"""
import csv
import json
import os
import subprocess
import sys
import time


def per_launch(path, needle):
    rows = [r for r in csv.DictReader(open(path)) if needle in r["Kernel_Name"]]
    assert rows, (path, needle)
    name = rows[0]["Kernel_Name"]
    return sum(float(r["Counter_Value"]) for r in rows) / len(rows), len(rows), name


def main():
    d, tag = sys.argv[1], sys.argv[2]
    fetch, nf, kname = per_launch(os.path.join(d, "%s_bench_FETCH_SIZE_counter_collection.csv" % tag), "rows_batch_kernel")
    write, nw, _ = per_launch(os.path.join(d, "%s_bench_WRITE_SIZE_counter_collection.csv" % tag), "rows_batch_kernel")
    cal = {"fetch_correction": 2.0, "write_correction": 1.0, "note": "calibration kernel not run: the guide's gfx950 factor is used"}
    cf = os.path.join(d, "%s_calib_FETCH_SIZE_counter_collection.csv" % tag)
    cw = os.path.join(d, "%s_calib_WRITE_SIZE_counter_collection.csv" % tag)
    if os.path.exists(cf) and os.path.exists(cw):
        f, _, cname = per_launch(cf, "k_copy")
        w, _, _ = per_launch(cw, "k_copy")
        known = 5.12e9
        cal = {"kernel": "k_copy (experiments/gather_variants.hip): streams exactly 5.12e9 B in and 5.12e9 B out",
               "FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "fetch_correction": round(known / (f * 1024), 3),
               "write_correction": round(known / (w * 1024), 3)}
    rd = fetch * 1024 * cal["fetch_correction"]
    wr = write * 1024 * cal["write_correction"]
    algo = 10_000_000 * (8 + 512 + 512)
    try:
        sha = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        sha = os.environ.get("WM_COMMIT", "worktree")   # no .git on the GPU box: the caller passes the commit it snapshots
    out = {
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, kernel trace only) around `python bench.py "
                  "--no-cpu-baseline --steps 10 --warmup 2`, raw CSVs: profiles/%s_pmc/" % tag,
        "collected": "round %s, %s" % (tag, time.strftime("%Y-%m-%d")), "commit": sha,
        "kernel": kname, "launches_averaged": {"FETCH_SIZE": nf, "WRITE_SIZE": nw},
        "units": "counter values are KiB",
        "calibration": cal,
        "FETCH_SIZE_KiB_per_launch": fetch, "WRITE_SIZE_KiB_per_launch": write,
        "gather_read_bytes_per_launch": rd, "gather_write_bytes_per_launch": wr,
        "gather_hbm_bytes_per_launch": int(rd + wr),
        "algorithmic_bytes_per_launch": algo,
        "traffic_over_algorithmic": round((rd + wr) / algo, 4),
    }
    json.dump(out, open(os.path.join(d, "pmc_traffic.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
