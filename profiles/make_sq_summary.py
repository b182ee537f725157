#!/usr/bin/env python3
"""Per-kernel SQ counters from one rocprofv3 --pmc pass (8 SQ slots on gfx950, MI355X_MICROARCH.md "rocprofv3 PMC
slots"): MFMA-pipe busy, VALU-active, parked and issue-stalled wave cycles as fractions of the kernel's wave cycles,
averaged over the last n launches of each kernel.

    python profiles/make_sq_summary.py <counter_collection.csv> <last_n> <workload>
"""
import collections
import csv
import json
import sys


def main():
    path, last, workload = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {"workload": workload, "launches_averaged": last, "kernels": {}}
    # every kernel of the timed step that the pass instrumented, matched by PREFIX (template arguments and kernel
    # generations change from round to round: a whitelist of exact names silently dropped the round-4 kernels)
    prefixes = ("k_stream", "k_fc1_agg", "k_embed_out", "k_out_gru", "k_gru", "k_build_messages", "k_last_pos", "k_pruned_topk",
                "k_project_rows", "k_affinity", "k_deps", "k_own", "k_reserve", "k_hot_select", "k_plan", "k_prepass")
    for k in sorted(agg):
        if not k.startswith(prefixes):
            continue
        c = {n: sum(v[-last:]) / len(v[-last:]) for n, v in agg[k].items()}
        wc, busy = c.get("SQ_WAVE_CYCLES", 0.0), c.get("SQ_BUSY_CYCLES", 0.0)
        # SQ_BUSY_CYCLES is accumulated per shader engine (32 on MI355X: the value is 32 x kernel duration x clock),
        # SQ_VALU_MFMA_BUSY_CYCLES per SIMD (1024): MFMA-pipe busy fraction = MFMA_BUSY / (BUSY / 32 * 1024)
        out["kernels"][k] = {
            "counters": c,
            "mfma_pipe_busy_frac": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (busy * 32.0) if busy else None,
            "valu_active_frac_of_wave_cycles": c.get("SQ_ACTIVE_INST_VALU", 0.0) / wc if wc else None,
            "parked_frac_of_wave_cycles": c.get("SQ_WAIT_ANY", 0.0) / wc if wc else None,
            "issue_stalled_frac_of_wave_cycles": c.get("SQ_WAIT_INST_ANY", 0.0) / wc if wc else None,
            "mfma_mops_f32": c.get("SQ_INSTS_VALU_MFMA_MOPS_F32"),
        }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
