#!/usr/bin/env python3
"""Raw addresses of a native stack trace -> library + offset (+ the nearest exported symbol), from the /proc/self/maps of
the process that crashed (bench.py writes it when ZT_DUMP_MAPS is set).
    python tools/symbolise.py <maps file> <stderr log with '@ 0x...' frames>
The libraries are looked up at the paths in the maps file (the GPU box and the build container share one image)."""
import bisect
import os
import re
import subprocess
import sys


def main():
    maps, log = sys.argv[1], sys.argv[2]
    regions = []
    for ln in open(maps):
        f = ln.split()
        if len(f) < 6 or not f[5].startswith("/"):
            continue
        lo, hi = [int(x, 16) for x in f[0].split("-")]
        regions.append((lo, hi, int(f[2], 16), f[5]))
    regions.sort()
    starts = [r[0] for r in regions]
    syms = {}

    def symbols(path):
        if path not in syms:
            tab = []
            if os.path.exists(path):
                for tool in (["nm", "-D", "--defined-only", "-C"], ["nm", "--defined-only", "-C"]):
                    out = subprocess.run(tool + [path], capture_output=True, text=True).stdout
                    for l in out.splitlines():
                        p = l.split(None, 2)
                        if len(p) == 3 and p[1] in "TtWw":
                            tab.append((int(p[0], 16), p[2]))
            syms[path] = sorted(set(tab))
        return syms[path]

    # the lowest mapping of a library = its load base
    base = {}
    for lo, hi, off, path in regions:
        base[path] = min(base.get(path, lo), lo - off)
    for ln in open(log, errors="replace"):
        m = re.search(r"@\s+(0x[0-9a-f]+)\s+(.*)", ln)
        if not m:
            continue
        a = int(m.group(1), 16)
        i = bisect.bisect_right(starts, a) - 1
        if i < 0 or not (regions[i][0] <= a < regions[i][1]):
            print("%#x  ?  %s" % (a, m.group(2)))
            continue
        path = regions[i][3]
        rel = a - base[path]
        tab = symbols(path)
        j = bisect.bisect_right([t[0] for t in tab], rel) - 1
        near = "%s+%#x" % (tab[j][1][:90], rel - tab[j][0]) if j >= 0 else "?"
        print("%#x  %s+%#x  %s" % (a, os.path.basename(path), rel, near))


if __name__ == "__main__":
    main()
