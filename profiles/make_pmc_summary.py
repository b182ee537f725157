#!/usr/bin/env python3
"""Per-kernel HBM traffic from two separate rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE cannot
share a pass on gfx950).  Units and correction as MI355X_MICROARCH.md "HBM" prescribes: the counters
are in KB; FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950 (doubled here; 8-byte-per-lane
accesses are uncalibrated, so the corrected read figure is an upper bound for the T-PPR kernel);
WRITE_SIZE is exact.

    python profiles/make_pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <last_n> <workload>
                                        [<bench line of the profiled command (json)> [<commit>]]

The summary records what it was measured on -- commit, prefill, batches per T-PPR launch, T-PPR CUs, edges per k_stream
launch -- so that bench.py merges its `traffic` only into a line of the same launch shape.
"""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def per_kernel(path, counter, last):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        agg[name].append(float(r["Counter_Value"]))
    return {k: sum(v[-last:]) / len(v[-last:]) for k, v in agg.items()}


def _csrc_sha():
    try:
        from zebra_amd.build import csrc_sha16
        return csrc_sha16()
    except Exception:
        return None


def main():
    fetch, write, last, workload = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    f, w = per_kernel(fetch, "FETCH_SIZE", last), per_kernel(write, "WRITE_SIZE", last)
    out = {"workload": workload, "launches_averaged": last, "unit": "bytes per launch", "kernels": {}}
    if len(sys.argv) > 5:
        line = [l for l in open(sys.argv[5]).read().splitlines() if l.startswith('{"metric"')]
        if line:
            d = json.loads(line[-1])
            cfg = d.get("config", {})
            out["measured_on"] = {"commit": sys.argv[6] if len(sys.argv) > 6 else None, "workload_line": cfg.get("workload"),
                                  "tppr_launch_group": cfg.get("tppr_launch_group"), "tppr_cus": cfg.get("tppr_cus"),
                                  "steps": d.get("steps"), "csrc_sha16": _csrc_sha(),
                                  "edges_per_k_stream_launch": (cfg.get("tppr_launch_group") or 1) * (cfg.get("global_batch") or 0)}
    # kernel-name prefix -> bench.py's name (template arguments vary with the workload)
    prefixes = [("k_stream", "tppr_stream"), ("k_fc1_agg_reg", "fc1_agg"), ("k_fc1_agg_wide", "fc1_agg"), ("k_fc1_agg_d100", "fc1_agg"),
                ("k_fc1_agg<true>", "fc1_agg_generic"), ("k_fc1_agg<false>", "fc1_agg_full"), ("k_embed_out", "embed_out"),
                ("k_out_gru", "gru_update"), ("k_gru_split", "gru_update"), ("k_gru", "gru_update"), ("k_build_messages", "store_messages"), ("k_pruned_topk", "pruned_topk"),
                ("k_deps", "tppr_prepass"), ("k_prepass_fused", "tppr_prepass"), ("k_project_rows", "project_rows"),
                ("k_affinity", "score")]
    for pre, n in prefixes:                        # (the first prefix in the list that names a kernel of this run wins its slot)
        for k in sorted(set(f) | set(w)):
            if k.startswith(pre) and n not in out["kernels"]:
                fr, wr = f.get(k, 0.0) * 1024, w.get(k, 0.0) * 1024
                out["kernels"][n] = {"kernel": k, "fetch_raw": fr, "write": wr, "traffic": 2 * fr + wr}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
