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


# experiments/zipf_stats.py: Zipf(1.05), 10 M ids per rank on W x 125 M rows — per SENDER, the copies and the distinct ids that go
# to ONE owner (the pair that binds the step is the largest one)
ZIPF = {
    "zipf": {2: dict(distinct=4851758, hottest=525703, max_copies=5156954, max_distinct=2425957),
             4: dict(distinct=4880987, hottest=525979, max_copies=2752569, max_distinct=1221095),
             8: dict(distinct=4898766, hottest=525979, max_copies=1584769, max_distinct=612984)},
    "zipf_clustered": {2: dict(distinct=4851673, hottest=525703, max_copies=8298900, max_distinct=3175880),
                       4: dict(distinct=4880850, hottest=525979, max_copies=7466266, max_distinct=2369505),
                       8: dict(distinct=4898680, hottest=525979, max_copies=7063196, max_distinct=1976741)},
}
ADD_NS = 2.4    # per row of the ordered fold of a very long run through its dense copy (round 6: 5.4-5.8 cycles per row at 2.4 GHz,
                # profiles/r06_fold5_harness_slices.txt, r06_grad_timeline_zipf_dense_third.txt; round 5's chain figure was 2.62)


def link_ms(rows, row_bytes, gbps):
    return rows * (row_bytes + 8) / (gbps * 1e9) * 1e3


def predicted_zipf(n, variant, row_bytes=512):
    """DESIGN.md section 4, skewed batches. The step is bound by the LARGEST ordered pair (one xGMI link per pair).
    gather: the requester de-duplicates (decided from the duplicate estimate in the counts exchange) -> distinct ids travel;
    gradient apply: every copy travels (reference, and the ordered fp32 fold) or, with a fold that is free of the reference's order,
    one partial row per distinct id and sender (round 6: combined_gradient_apply); the ordered fold of the hottest id at its
    owner is a dependent chain of n x hottest adds."""
    z = ZIPF[variant][n]
    rep = {"distinct_per_rank": z["distinct"], "copies_of_hottest_id_per_rank": z["hottest"]}
    for name, rows in (("every_copy", z["max_copies"]), ("distinct_only", z["max_distinct"])):
        rep["largest_pair_rows_" + name] = rows
        rep["largest_pair_bytes_" + name] = rows * (row_bytes + 8)
        rep["link_bound_ms_%s_at_76.8" % name] = round(link_ms(rows, row_bytes, 76.8), 2)
        rep["link_bound_ms_%s_at_153.6" % name] = round(link_ms(rows, row_bytes, 153.6), 2)
    rep["link_bytes_ratio_every_copy_over_distinct"] = round(z["max_copies"] / z["max_distinct"], 2)
    rep["ordered_fold_chain_ms_at_the_hot_ids_owner"] = round(n * z["hottest"] * ADD_NS * 1e-6, 2)
    # gradient apply, whole step: sender-side work in front of the links + the link-bound exchange + the owner's step behind it
    # (one-GPU measurements: sender combination ~2 ms per 10 M rows incl. its id sort, owner step over n x distinct partial rows
    # ~0.5 ms per M rows; the uncombined owner step overlaps its id sort with the tail of the exchange)
    owner_ms = 0.5e-6 * n * z["max_distinct"] * 1.0
    rep["grad_apply_step_ms_combined_at_76.8"] = round(2.0 + link_ms(z["max_distinct"], row_bytes, 76.8) + owner_ms, 1)
    rep["grad_apply_step_ms_every_copy_tree_at_76.8"] = round(0.6 + link_ms(z["max_copies"], row_bytes, 76.8) + 2.4, 1)
    rep["grad_apply_step_ms_every_copy_ordered_at_76.8"] = round(
        0.6 + link_ms(z["max_copies"], row_bytes, 76.8) + max(2.4, rep["ordered_fold_chain_ms_at_the_hot_ids_owner"]), 1)
    return rep


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
                                      "config", "exchange", "roofline", "c3_zipf", "stability", "side_errors", "grad_route") if k in r}
        lines[name] = keep
    rep = {"ranks": n, "dry_run": dry, "failed_step": failed or None,
           "note": "dry run: toy sizes over gloo on shared devices — checks the script, says nothing about the links" if dry else
                   "RCCL over xGMI, BASELINE sizes (125 M rows and 10 M ids per rank)",
           "predictions_uniform": {str(k): predicted(k) for k in (1, 2, 4, 8) if k <= max(n, 1)},
           "predictions_zipf_hashed": {str(k): predicted_zipf(k, "zipf") for k in (2, 4, 8) if k <= n},
           "predictions_zipf_clustered": {str(k): predicted_zipf(k, "zipf_clustered") for k in (2, 4, 8) if k <= n},
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
    # the skewed legs against THEIR predictions: C3 Zipf (gather, requester de-duplication) and C4 (gradient apply, with and
    # without the sender-side combination; alltoallv_bytes_per_step is the library's own byte counter when the line carries it)
    zc = {}
    for k in (2, 4, 8):
        for variant, leg in (("zipf", "c3_zipf_n%d"), ("zipf_clustered", "c3_zipf_clustered_n%d")):
            m = lines.get(leg % k)
            if m and "ms_per_step" in m and not dry:
                p = predicted_zipf(k, variant)
                zc[leg % k] = {"measured_ms": m["ms_per_step"], "link_bound_ms_distinct_only_at_76.8": p["link_bound_ms_distinct_only_at_76.8"],
                               "link_bound_ms_distinct_only_at_153.6": p["link_bound_ms_distinct_only_at_153.6"],
                               "link_bound_ms_every_copy_at_76.8": p["link_bound_ms_every_copy_at_76.8"]}
    if n in ZIPF["zipf"] and not dry:
        p = predicted_zipf(n, "zipf")
        for leg, key in (("c4_grad_apply_f16_n%d", "grad_apply_step_ms_combined_at_76.8"),
                         ("c4_grad_apply_f16_every_copy_n%d", "grad_apply_step_ms_every_copy_tree_at_76.8"),
                         ("c4_grad_apply_f32_tree_n%d", "grad_apply_step_ms_combined_at_76.8"),
                         ("c4_grad_apply_f32_n%d", "grad_apply_step_ms_every_copy_ordered_at_76.8")):
            m = lines.get(leg % n)
            if m and "ms_per_step" in m:
                zc[leg % n] = {"measured_ms": m["ms_per_step"], "predicted_ms": p[key], "prediction": key,
                               "alltoallv_bytes_per_step": (m.get("exchange") or {}).get("alltoallv_bytes_per_step")}
    rep["skewed_vs_prediction"] = zc
    json.dump(rep, open(os.path.join(out, "first_contact.json"), "w"), indent=1)
    print("first_contact_report: %d bench lines -> %s" % (len(lines), os.path.join(out, "first_contact.json")))
    return 0


if __name__ == "__main__":
    sys.exit(main())
