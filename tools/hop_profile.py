import sys, ctypes as C
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from zebra_amd import _capi
_capi.LIB_PATH = '/root/repo/tools/out/libzebra_stamp.so'
from zebra_amd import tppr, synth
lib = _capi.lib()
wl = synth.WORKLOADS["c5"]; B = 4096; NB = int(sys.argv[1]) if len(sys.argv) > 1 else 60
src, dst, ts, eidx = synth.power_law_stream(wl["n_nodes"], NB * B, seed=2020)
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
print("avg k_stream us:", 1e3 * ms.value / n.value)
st = np.zeros((B, 4), np.int64)
lib.zt_debug_stamps(st.ctypes.data_as(C.c_void_p), C.c_int(B))
t = (st - st[:, 0].min()) * 0.01
s0 = (NB - 1) * B
u = src[s0:s0 + B]; v = dst[s0:s0 + B]
print("kernel span (last batch, model 0 tasks): %.1f us" % (t[:, 3].max() - t[:, 0].min()))
hub = np.bincount(np.concatenate([u, v])).argmax()
idx = np.where((u == hub) | (v == hub))[0]
print("hub", hub, "edges", len(idx))
x1 = t[idx, 2]; rr = t[idx, 1]; dq = t[idx, 0]; en = t[idx, 3]
print("first hub edge rows-ready at %.1f, last hub edge end at %.1f" % (rr[0], en[-1]))
hop = np.diff(x1)
print("hub hop (x1 stored -> next x1 stored) percentiles 10/50/90/99:", np.percentile(hop, [10, 50, 90, 99]).round(2), "mean %.2f" % hop.mean())
print("handoff (x1 stored i -> rows ready i+1):", np.percentile(rr[1:] - x1[:-1], [10, 50, 90, 99]).round(2))
print("merge (rows ready -> x1 stored):", np.percentile(x1 - rr, [10, 50, 90, 99]).round(2))
print("dequeued after predecessor finished (late dequeue) count:", int(np.sum(dq[1:] > x1[:-1])))
late = np.maximum(dq[1:] - x1[:-1], 0)
print("late-dequeue delay sum %.1f us" % late.sum())
s2 = np.zeros((B, 8), np.int64)
lib.zt_debug_stamps2(s2.ctypes.data_as(C.c_void_p), C.c_int(B))
ph = (s2[idx, 1:6] - s2[idx, 0:5]) * 0.01
print("hub-edge first-merge phases median (r1->LDS, search, newkey, topk, readback):", np.median(ph, axis=0).round(2))
tk = ph[:, 3]
print("topk duration percentiles 10/25/50/75/90:", np.percentile(tk, [10, 25, 50, 75, 90]).round(2))
allph = (s2[:, 1:6] - s2[:, 0:5]) * 0.01
ok = (s2[:, 5] > 0)
print("ALL edges: topk percentiles:", np.percentile(allph[ok, 3], [10, 25, 50, 75, 90]).round(2), " frac topk>2us: %.2f" % np.mean(allph[ok, 3] > 2.0))
pp = np.zeros(16, np.int32); lib.zt_debug_paths(pp.ctypes.data_as(C.c_void_p)); print("topk paths [rank, lds, seq, reg, ties]:", pp)
# set chain of the hub: stage-1 publication times (stamp 7) and where a hop spends its turn
pub = s2[idx, 7] * 0.01; seen = s2[idx, 6] * 0.01; st = s2[idx, 0] * 0.01; rk = s2[idx, 4] * 0.01
okm = (pub[1:] > 0) & (pub[:-1] > 0) & (seen[1:] > 0)
d = lambda a: np.percentile(a[okm], [10, 50, 90]).round(2)
print("set chain: publish(t) - publish(t-1):", d(pub[1:] - pub[:-1]), "mean %.2f" % (pub[1:] - pub[:-1])[okm].mean())
print("  publish(t-1) -> seen(t):", d(seen[1:] - pub[:-1]), " seen -> merge start:", d(st[1:] - seen[1:]),
      " merge start -> rank pass end:", d(rk[1:] - st[1:]), " rank pass end -> publish:", d(pub[1:] - rk[1:]))

# ---- slow hops: what was the hub waiting for? ----
hop_full = np.diff(x1)
slow = np.where(hop_full > 6.0)[0] + 1           # hop index (in idx) whose arrival was late
print("slow hops (> 6 us):", len(slow), "of", len(idx), " extra time %.1f us" % (hop_full[hop_full > 6.0] - np.median(hop_full)).sum())
last_writer = {}
deg = np.bincount(np.concatenate([u, v]), minlength=1)
for i in range(B):
    pass
prev_touch = {}
writer_of = np.full((B, 2), -1)
for i in range(B):
    for r, x in enumerate((u[i], v[i])):
        writer_of[i, r] = prev_touch.get(x, -1)
    prev_touch[u[i]] = i; prev_touch[v[i]] = i
for s_i in slow[:12]:
    e = idx[s_i]
    partner = v[e] if u[e] == hub else u[e]
    r = 1 if u[e] == hub else 0
    j = writer_of[e, r]
    msg = "edge %d partner %d (deg %d in batch)" % (e, partner, deg[partner] if partner < len(deg) else -1)
    if j >= 0:
        msg += "  last writer edge %d: deq %.1f rows %.1f x1 %.1f end %.1f" % (j, t[j, 0], t[j, 1], t[j, 2], t[j, 3])
    msg += "  | me: deq %.1f rows %.1f x1 %.1f ; prev hub x1 %.1f" % (t[e, 0], t[e, 1], t[e, 2], t[idx[s_i - 1], 2])
    print(msg)

# ---- the chain's first hops, phase by phase (stamps: 6 seen, 0 merge start, 1 matched, 2 laid out, 4 ranked, 7 published) ----
print("first hops: [publish(t-1)->seen, seen->merge, scales+match, layout, rank pass, ->publish] us")
for q in range(1, 14):
    e, ep = idx[q], idx[q - 1]
    a = s2[e] * 0.01
    print("  hop %2d edge %4d: %5.2f %5.2f %5.2f %5.2f %5.2f %5.2f   total %.2f" % (
        q, e, a[6] - s2[ep, 7] * 0.01, a[0] - a[6], a[1] - a[0], a[2] - a[1], a[4] - a[3], a[7] - a[4], (s2[e, 7] - s2[ep, 7]) * 0.01))
