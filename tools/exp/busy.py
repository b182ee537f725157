# who is the bottleneck?  From a rocprofv3 kernel trace of a bench run: over the last n steps, the share of the time the
# T-PPR update (k_stream) is running, the share the main stream's kernels are running, and the gaps between launches.
#   python3 tools/exp/busy.py kt_kernel_trace.csv 200
import csv, sys
import numpy as np
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2])
agg = [r for r in rows if "k_fc1_agg" in r["Kernel_Name"]][-n:]
t0, t1 = int(agg[0]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
def sel(pred):
    return [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if pred(r["Kernel_Name"]) and int(r["End_Timestamp"]) > t0]
ks = sel(lambda k: "k_stream" in k)
main = sel(lambda k: any(x in k for x in ("k_fc1_agg", "k_embed_out", "k_gru", "k_select_flagged")))
msg = sel(lambda k: any(x in k for x in ("k_last_pos", "k_build_messages")))
span = (t1 - t0) / 1e3
print("region %.1f us, %.1f us per step" % (span, span / n))
for name, iv in (("k_stream", ks), ("main stream (fc1_agg, embed_out, select, gru)", main), ("message kernels", msg)):
    busy = sum(e - max(s, t0) for s, e in iv) / 1e3
    gaps = np.array([iv[i + 1][0] - iv[i][1] for i in range(len(iv) - 1)]) / 1e3
    print("%-48s busy %.1f us (%.0f %% of the region), %d launches, gaps between launches: median %.1f mean %.1f max %.1f us" % (
        name, busy, 100 * busy / span, len(iv), np.median(gaps), gaps.mean(), gaps.max()))
# per step: from the start of one fc1_agg to the start of the next
st = np.array([int(r["Start_Timestamp"]) for r in agg]) / 1e3
d = np.diff(st)
print("fc1_agg start to start: median %.1f mean %.1f us; main-stream work per step (sum of its kernels) %.1f us" % (
    np.median(d), d.mean(), sum(e - s for s, e in main) / 1e3 / n))
