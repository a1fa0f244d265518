#!/usr/bin/env python3
"""first_contact.json: the bench lines scripts/first_contact.sh collected, next to DESIGN.md section 4's predictions.
usage: first_contact_report.py <out dir> <N> <dry 0|1> <failed step or ''>"""
import glob
import json
import os
import sys


def predicted(n, ids_per_rank=10_000_000, row_bytes=512):
    """uniform ids, one xGMI link per GPU pair, ids / n x (row + 8 id bytes) per ordered pair per step;
    link-bound time at 76.8 GB/s per direction (and at 153.6, if the quoted link figure is per direction)"""
    if n == 1:
        return {"step_ms": 1.71, "aggregate_GBps_out": 3000.0, "source": "measured, round 4"}
    pair = ids_per_rank / n * (row_bytes + 8)
    lo, hi = pair / 76.8e9 * 1e3, pair / 153.6e9 * 1e3
    local = 1.8   # owner gather + reorder of a rank's 10 M rows, overlapped with the links except for the first / last chunk
    return {"bytes_per_ordered_pair": pair, "link_bound_ms_at_76.8": round(lo, 2), "link_bound_ms_at_153.6": round(hi, 2),
            "step_ms_at_76.8": round(lo + local / 4, 2), "step_ms_at_153.6": round(hi + local / 4, 2),
            "aggregate_GBps_out_at_76.8": round(n * ids_per_rank * row_bytes / ((lo + local / 4) * 1e-3) / 1e9, 0),
            "aggregate_GBps_out_at_153.6": round(n * ids_per_rank * row_bytes / ((hi + local / 4) * 1e-3) / 1e9, 0)}


def main():
    out, n, dry, failed = sys.argv[1], int(sys.argv[2]), sys.argv[3] == "1", sys.argv[4]
    lines = {}
    for f in sorted(glob.glob(os.path.join(out, "*.json"))):
        name = os.path.basename(f)[:-5]
        if name == "first_contact":
            continue
        try:
            r = json.loads(open(f).read().strip().splitlines()[-1])
        except Exception as ex:  # noqa
            lines[name] = {"error": "no bench line (%s)" % ex}
            continue
        keep = {k: r.get(k) for k in ("metric", "value", "unit", "n_gpus", "rccl_ranks", "transport", "ms_per_step", "mlookups_per_s",
                                      "config", "exchange", "roofline", "c3_zipf", "stability", "side_errors") if k in r}
        lines[name] = keep
    rep = {"ranks": n, "dry_run": dry, "failed_step": failed or None,
           "note": "dry run: toy sizes over gloo on shared devices — checks the script, says nothing about the links" if dry else
                   "RCCL over xGMI, BASELINE sizes (125 M rows and 10 M ids per rank)",
           "predictions_uniform": {str(k): predicted(k) for k in (1, 2, 4, 8) if k <= max(n, 1)},
           "measured": lines}
    # the comparison the first SCALE record is held against
    cmp_ = {}
    for k in (2, 4, 8):
        m = lines.get("c3_uniform_n%d" % k)
        if m and "ms_per_step" in m and not dry:
            p = predicted(k)
            cmp_[str(k)] = {"measured_ms": m["ms_per_step"], "predicted_ms_at_76.8": p["step_ms_at_76.8"],
                            "predicted_ms_at_153.6": p["step_ms_at_153.6"],
                            "reads_the_link_figure_as": "76.8 GB/s per direction" if abs(m["ms_per_step"] - p["step_ms_at_76.8"]) <
                            abs(m["ms_per_step"] - p["step_ms_at_153.6"]) else "153.6 GB/s per direction"}
    rep["uniform_vs_prediction"] = cmp_
    json.dump(rep, open(os.path.join(out, "first_contact.json"), "w"), indent=1)
    print("first_contact_report: %d bench lines -> %s" % (len(lines), os.path.join(out, "first_contact.json")))
    return 0


if __name__ == "__main__":
    sys.exit(main())
