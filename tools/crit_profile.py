# Length of the critical section of a hub hop in core clocks (register-held readings, -DZT_CRIT build).
import sys, ctypes as C
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from zebra_amd import _capi
_capi.LIB_PATH = '/root/repo/tools/out/libzebra_crit.so'
from zebra_amd import tppr, synth
lib = _capi.lib()
WL = sys.argv[2] if len(sys.argv) > 2 else "c5"        # crit_profile.py [batches] [workload] [batches per launch]
G = int(sys.argv[3]) if len(sys.argv) > 3 else 1
wl = synth.WORKLOADS[WL]; B = wl["bs"] * G; NB = int(sys.argv[1]) if len(sys.argv) > 1 else 200
src, dst, ts, eidx = synth.power_law_stream(wl["n_nodes"], NB * B, bipartite=wl["bipartite"], seed=2020, perm_seed=7)
neg = synth.negatives(dst, len(src), seed=2021)
f = tppr.tppr_finder(wl["n_nodes"] + 1, 20, 2, [0.1, 0.1], [0.5, 0.95])
d = torch.device('cuda')
sd, dd, nd = [torch.from_numpy(x).to(d) for x in (src, dst, neg)]
td, ed = torch.from_numpy(ts).to(d), torch.from_numpy(eidx).to(d)
lib.zt_profile_reset(); lib.zt_profile_enable(1)
for b in range(NB):
    s, e = b * B, (b + 1) * B
    f.stream_device(torch.cat([sd[s:e], dd[s:e], nd[s:e]]), td[s:e], ed[s:e], 3, True, -1, check_status=False)
f.check_status()
n, ms = C.c_int64(), C.c_double(); lib.zt_profile_read(b"tppr_stream", C.byref(n), C.byref(ms))
print("avg k_stream us: %.1f" % (1e3 * ms.value / n.value))
c2 = np.zeros((2 * B, 16), np.int64)
lib.zt_debug_crit(c2.ctypes.data_as(C.c_void_p), C.c_int(2 * B))
c = c2[:B]
s0 = (NB - 1) * B
u, v = src[s0:s0 + B], dst[s0:s0 + B]
hub = np.bincount(np.concatenate([u, v])).argmax()
idx = np.where((u == hub) | (v == hub))[0]
cc = c[idx]
ok = (cc[:, 0] > 0) & (cc[:, 3] > 0)
cc = cc[ok]
p = lambda a: np.percentile(a, [10, 50, 90]).round(0)
print("hub edges %d, split hops with readings %d" % (len(idx), len(cc)))
print("core clocks 10/50/90 %%: row arrived -> front half done %s ; -> ready to publish %s ; publication %s ; whole section %s" % (
    p(cc[:, 1] - cc[:, 0]), p(cc[:, 2] - cc[:, 1]), p(cc[:, 3] - cc[:, 2]), p(cc[:, 3] - cc[:, 0])))
print("inside the front half: arrival -> merge_front entered %s ; scales + matching + new key %s ; layout %s ; rank pass %s ; return %s" % (
    p(cc[:, 4] - cc[:, 0]), p(cc[:, 5] - cc[:, 4]), p(cc[:, 6] - cc[:, 5]), p(cc[:, 7] - cc[:, 6]), p(cc[:, 1] - cc[:, 7])))
print("arrival -> row read from LDS %s ; -> split path entered %s ; -> front half called %s ; -> inside %s" % (p(cc[:, 8] - cc[:, 0]), p(cc[:, 9] - cc[:, 8]), p(cc[:, 10] - cc[:, 9]), p(cc[:, 4] - cc[:, 10])))
print("front half taken: fast %d | partner side not prepared %d | norm differs %d | key match %d | new key / NaN in the hub row %d | row not sorted %d | clash late %d" % tuple([int((cc[:, 11] == q).sum()) for q in (0, 1, 2, 3, 4, 5, 6)]))
print("front half done -> settled / redo checks %s ; -> ring slot free %s ; -> slots assigned %s ; -> set written %s" % (p(cc[:, 12] - cc[:, 1]), p(cc[:, 13] - cc[:, 12]), p(cc[:, 14] - cc[:, 13]), p(cc[:, 2] - cc[:, 14])))
arr = c[idx][:, 0]; pub = c[idx][:, 3]
g = (arr[1:] > 0) & (pub[:-1] > 0)
print("publication(t-1) -> row arrived(t) [clocks of two different waves, same CU]: %s" % p((arr[1:] - pub[:-1])[g]))
print("publication(t) - publication(t-1): %s   mean %.0f" % (p(np.diff(pub)[(pub[1:] > 0) & (pub[:-1] > 0)]), np.diff(pub)[(pub[1:] > 0) & (pub[:-1] > 0)].mean()))
d = np.diff(pub); okd = (pub[1:] > 0) & (pub[:-1] > 0)
q = len(d) // 4
print("hop to hop by quarter of the chain (median, mean): " + " | ".join("%.0f %.0f" % (np.median(d[a:a + q][okd[a:a + q]]), d[a:a + q][okd[a:a + q]].mean()) for a in range(0, 4 * q, q)))
sec = (c[idx][:, 3] - c[idx][:, 0]); oks = (c[idx][:, 3] > 0) & (c[idx][:, 0] > 0)
print("critical section by quarter (median): " + " | ".join("%.0f" % np.median(sec[a:a + q][oks[a:a + q]]) for a in range(0, 4 * q, q)))
lean = c[idx][:, 7] == 1
cl = c[idx][lean]
if len(cl):
    print("lean hops %d of %d: preparation (dequeue -> ready to wait) %s ; slack (ready -> row arrived; small = the wave was late) %s ; critical section %s ; tail (published -> hop done) %s ; whole hop %s" % (
        len(cl), len(idx), p(cl[:, 5] - cl[:, 4]), p(cl[:, 0] - cl[:, 5]), p(cl[:, 3] - cl[:, 0]), p(cl[:, 6] - cl[:, 3]), p(cl[:, 6] - cl[:, 4])))
    print("  mean: preparation %.0f, critical %.0f, tail %.0f, whole %.0f; late arrivals (slack < 300): %d" % ((cl[:, 5] - cl[:, 4]).mean(), (cl[:, 3] - cl[:, 0]).mean(), (cl[:, 6] - cl[:, 3]).mean(), (cl[:, 6] - cl[:, 4]).mean(), int(((cl[:, 0] - cl[:, 5]) < 300).sum())))
ci = c[idx]
dd = np.diff(ci[:, 3]); slow = np.where(dd > 4000)[0] + 1
print("slow hops (> 4000 clocks after their predecessor's publication): %d, sum of their excess over the median %.0f clocks" % (len(slow), (dd[slow - 1] - np.median(dd)).sum()))
for t in slow[:40]:
    r = ci[t]
    print("  t=%3d lean=%d fallback=%d  hop %6d  hand-off %6d  critical %6d  ready->arrival %7d  prep %6d" % (t, r[7] == 1, r[11], dd[t - 1], r[0] - ci[t - 1][3], r[3] - r[0], r[0] - r[5], r[5] - r[4]))
for mo in (0, 1):
    cm = c2[mo * B:(mo + 1) * B][idx]
    okm = cm[:, 3] > 0
    st = cm[okm][:, 4].min(); en = cm[okm][:, 3].max()
    dm = np.diff(cm[:, 3]); okd2 = (cm[1:, 3] > 0) & (cm[:-1, 3] > 0)
    print("model %d: chain span (first stamped hop's start -> last publication) %.0f clocks = %.1f us at 2.35 GHz; hop to hop median %.0f mean %.0f; lean %d of %d" % (
        mo, en - st, (en - st) / 2350.0, np.median(dm[okd2]), dm[okd2].mean(), int((cm[:, 7] == 1).sum()), len(cm)))
rl = cl[cl[:, 8] > 0]
if len(rl):
    print("off-chain half of the %d lean hops that replay: published -> replay done %s ; -> previous hop's order there %s ; -> own order published %s ; -> hop done (stores) %s" % (
        len(rl), p(rl[:, 8] - rl[:, 3]), p(rl[:, 9] - rl[:, 8]), p(rl[:, 10] - rl[:, 9]), p(rl[:, 6] - rl[:, 10])))
cm = c[idx]; okm = (cm[:, 3] > 0) & (cm[:, 12] > 0)
if okm.any():
    t0k = cm[okm][:, 12].min(); first = cm[okm][:, 4].min(); lastp = cm[okm][:, 3].max()
    kern_clk = 1e3 * ms.value / n.value * 2350.0
    print("model 0: workgroup start -> first stamped hop starts %.0f clocks (%.1f us); -> last publication %.0f (%.1f us); average kernel %.1f us => %.1f us after the last publication (+ launch)" % (
        first - t0k, (first - t0k) / 2350.0, lastp - t0k, (lastp - t0k) / 2350.0, kern_clk / 2350.0, (kern_clk - (lastp - t0k)) / 2350.0))
if okm.any():
    print("         last partner task (model 0) done %.1f us after the last publication; last general wave left %.1f us after it" % (
        (c2[8191][1] - lastp) / 2350.0, (c2[8191][0] - lastp) / 2350.0))
lag = (cm[:, 14] - cm[:, 3])[(cm[:, 14] > cm[:, 3]) & (cm[:, 3] > 0) & (cm[:, 14] - cm[:, 3] < 2000000)] / 2350.0
if len(lag):
    qq = len(lag) // 4
    print("partner task done - publication of its hop (us), by quarter of the chain (median): " + " | ".join("%.1f" % np.median(lag[a:a + qq]) for a in range(0, 4 * qq, qq)) + "   last ten: " + " ".join("%.0f" % x for x in lag[-10:]))
part = np.where(u[idx] == hub, v[idx], u[idx])
dist = np.full(len(idx), 10 ** 6)
for t in range(len(idx)):
    w = np.where(part[max(0, t - 64):t] == part[t])[0]
    if len(w):
        dist[t] = t - (max(0, t - 64) + w[-1])
hop = np.concatenate([[0], np.diff(c[idx][:, 3])])
prep = c[idx][:, 5] - c[idx][:, 4]
rec = np.where(dist <= 64)[0]
print("hub hops whose partner was the partner of one of the previous 64 hub hops: %d of %d; their distance / preparation / hop time (clocks):" % (len(rec), len(idx)))
print("   " + " | ".join("%d %d %d" % (dist[t], prep[t], hop[t]) for t in rec[:40]))
ok2 = (hop > 0) & (hop < 400000)
print("   time of these hops above the median hop: %.0f clocks = %.1f %% of the chain's span" % ((hop[rec][ok2[rec]] - np.median(hop[ok2])).clip(min=0).sum(), 100 * (hop[rec][ok2[rec]] - np.median(hop[ok2])).clip(min=0).sum() / hop[ok2].sum()))
