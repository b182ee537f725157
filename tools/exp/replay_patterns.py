# How repetitive are the quicksort replays of a hub chain?  For the hub of C5's stream (bench.py's stream, 10 % prefill) walk two
# batches edge by edge on the CPU oracle; for every hub hop build the candidate list the reference's argsort sees (hub's row x
# scale_s1 in dictionary order, the partner's new keys x scale_s2, the new key: utils/util.py:514-559), take the RANK vector by
# list position (what topk_ties_reg replays) and count how many DISTINCT vectors the hops with ties produce -- a cache keyed by
# the rank vector would save a replay per repeat.   python tools/exp/replay_patterns.py [model] [batches]
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import pyoracle
from zebra_amd import synth
import bench

M_ID = int(sys.argv[1]) if len(sys.argv) > 1 else 0
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 2
wl = dict(synth.WORKLOADS["c5"]); bs, k = wl["bs"], wl["k"]
al, be = wl["alpha"], wl["beta"]
prefill = (wl["n_edges"] // 10) // bs
n = (prefill + NB) * bs
src, dst, neg, ts, eidx = bench.make_stream(wl, n, perm_seed=7)
f = pyoracle.TpprOracle(wl["n_nodes"] + 1, k, 2, al, be)
t0 = time.time()
CH = 1 << 16
for s in range(0, prefill * bs, CH):
    e = min(prefill * bs, s + CH)
    f._stream(np.concatenate([src[s:e], dst[s:e]]), ts[s:e], eidx[s:e], 2, False, -1)
print("prefill %.0f s" % (time.time() - t0), flush=True)
e0 = prefill * bs
hub = np.bincount(np.concatenate([src[e0:n], dst[e0:n]])).argmax()
alpha, beta = al[M_ID], be[M_ID]
pats, modes = {}, {"ranks": 0, "ties": 0, "straddle": 0, "noprune": 0}
seq = []
for i in range(e0, n):
    u, v = int(src[i]), int(dst[i])
    if (u == hub) != (v == hub):
        p = v if u == hub else u
        r = f.export_rows(M_ID, np.array([hub, p], np.int64))
        lh, lp = int(r["len"][0]), int(r["len"][1])
        nh, npn = float(r["norm"][0]), float(r["norm"][1])
        if nh != 0.0:
            nn = nh * beta + beta
            s1, s2 = nh / nn * beta, beta / nn * (1 - alpha)
            keys = [(int(r["eidx"][0][j]), int(r["node"][0][j]), float(r["ts"][0][j])) for j in range(lh)]
            vals = [float(r["w"][0][j]) * s1 for j in range(lh)]
            pos = {kk: j for j, kk in enumerate(keys)}
            if npn != 0.0:
                for j in range(lp):
                    kk = (int(r["eidx"][1][j]), int(r["node"][1][j]), float(r["ts"][1][j]))
                    w2 = float(r["w"][1][j]) * s2
                    if kk in pos: vals[pos[kk]] += w2
                    else: pos[kk] = len(vals); vals.append(w2)
            nk = (int(eidx[i]), p, float(ts[i]))
            wv = s2 * alpha if alpha != 0 else s2
            if nk in pos: vals[pos[nk]] = wv
            else: vals.append(wv)
            a = np.array(vals)
            if len(a) <= k: modes["noprune"] += 1
            else:
                lt = (a[None, :] < a[:, None]).sum(1)
                drop = len(a) - k
                kept = lt >= drop
                if kept.sum() == k and len(set(lt[kept])) == k: modes["ranks"] += 1
                else:
                    modes["ties" if kept.sum() == k else "straddle"] += 1
                    key = lt.astype(np.uint8).tobytes()
                    pats[key] = pats.get(key, 0) + 1
                    seq.append(key)
    f._stream(np.array([u, v, int(neg[i])], np.int32), ts[i:i + 1], eidx[i:i + 1], 3, False, -1)
tot = sum(pats.values())
print("model %d (beta %.2f): hub %d, hops by mode %s" % (M_ID, beta, hub, modes))
print("replays %d, distinct rank vectors %d (%.1f %% repeats); top counts %s" % (tot, len(pats), 100.0 * (tot - len(pats)) / max(1, tot), sorted(pats.values())[-8:]))
# hit rate of a small LRU cache
for cap in (4, 16, 64):
    lru, hits = [], 0
    for kk in seq:
        if kk in lru: hits += 1; lru.remove(kk)
        lru.append(kk)
        if len(lru) > cap: lru.pop(0)
    print("LRU of %d entries: %.1f %% hits" % (cap, 100.0 * hits / max(1, len(seq))))
