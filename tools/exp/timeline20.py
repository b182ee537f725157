# see timeline20.sh: per-queue timeline of the last `steps` batches of a bench run
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2])
agg = [i for i, r in enumerate(rows) if "k_fc1_agg" in r["Kernel_Name"]][-n:]
# the region starts after the synchronisation that follows the warm-up: the largest idle gap before the first timed aggregation
first = agg[0]
lo = max(0, first - 60)
best, start = -1, lo
end_so_far = max(int(r["End_Timestamp"]) for r in rows[:lo + 1])
for i in range(lo + 1, first + 1):
    gap = int(rows[i]["Start_Timestamp"]) - end_so_far
    if gap > best:
        best, start = gap, i
    end_so_far = max(end_so_far, int(rows[i]["End_Timestamp"]))
tz = int(rows[start]["Start_Timestamp"])
qs = {}
print("idle gap before the region: %.1f us; region = %d kernels" % (best / 1e3, len(rows) - start))
for r in rows[start:]:
    q = r.get("Queue_Id", "?")
    qs.setdefault(q, len(qs))
    s, e = (int(r["Start_Timestamp"]) - tz) / 1e3, (int(r["End_Timestamp"]) - tz) / 1e3
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:28]
    print("%9.1f -> %9.1f  %7.1f us  q%d  %s" % (s, e, e - s, qs[q], name))
print("region span %.1f us = %.4f ms/step" % ((int(rows[-1]["End_Timestamp"]) - tz) / 1e3, (int(rows[-1]["End_Timestamp"]) - tz) / 1e6 / n))
