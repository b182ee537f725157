# timeline of the timed region of a short bench run from a rocprofv3 kernel trace: T-PPR launches and aggregations
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2])                      # steps of the timed region
agg = [r for r in rows if "k_fc1_agg" in r["Kernel_Name"]][-n:]
t0 = None
ks = [r for r in rows if "k_stream" in r["Kernel_Name"] and int(r["End_Timestamp"]) > int(agg[0]["Start_Timestamp"]) - 3000000]
first_agg = int(agg[0]["Start_Timestamp"])
ks = [r for r in ks if int(r["Start_Timestamp"]) > first_agg - 1500000]
tz = int(ks[0]["Start_Timestamp"])
pre = [r for r in rows if tz - 400000 < int(r["Start_Timestamp"]) < tz]
if pre:
    tz = int(pre[0]["Start_Timestamp"])
print("region starts with", (pre[0] if pre else ks[0])["Kernel_Name"][:50])
for r in ks:
    print("  k_stream   %8.1f -> %8.1f  (%.1f us)" % ((int(r["Start_Timestamp"]) - tz) / 1e3, (int(r["End_Timestamp"]) - tz) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
print("  aggregations start at:", " ".join("%.0f" % ((int(r["Start_Timestamp"]) - tz) / 1e3) for r in agg))
print("  last kernel ends at %.1f (%s)" % ((int(rows[-1]["End_Timestamp"]) - tz) / 1e3, rows[-1]["Kernel_Name"][:30]))
