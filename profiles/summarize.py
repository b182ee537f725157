#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per kernel, calls / avg / total over
ALL launches and over the LAST n launches (the timed region of bench.py)."""
import csv
import sys
from collections import defaultdict


def main(path, last):
    rows = defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
            rows[name].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    tot_all = sum(sum(v) for v in rows.values())
    print("%-40s %7s %12s %12s | last %d launches: %12s" % ("kernel", "calls", "avg_us", "total_ms", last, "avg_us"))
    for name, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
        tail = v[-last:] if last else v
        print("%-40s %7d %12.1f %12.2f | %32.1f   (%.1f%% of kernel time)" % (
            name[:40], len(v), sum(v) / len(v) / 1e3, sum(v) / 1e6, sum(tail) / len(tail) / 1e3, 100.0 * sum(v) / tot_all))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 0)
