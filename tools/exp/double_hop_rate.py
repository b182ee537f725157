"""How often could TWO consecutive hops of a hub chain share one critical section?  (CPU only: the C oracle.)

The round-3 review's item 4: at k = 20 the hub's row (20) + two partners' sides (21 + 21) are 62 candidates -- one
64-lane bitonic network.  The pair (t, t+1) may be merged when
  (a) both hops are of the lean kind (the hub's row full, partner != hub, partner's row non-empty or empty alike),
  (b) the partners differ and neither is written by the other hop's edge (p1 != p2),
  (c) no key of one side is in another (hub / p1 / p2 pairwise disjoint),
  (d) the FINAL cut (top k of the 62 by their final weights) does not fall inside a run of equal weights.
This script walks the bench's stream on the oracle (prefill = 10 % like bench.py), stops at every edge of the
most-touched node of each launch, exports the hub's and the partner's rows and evaluates (a)-(d) for every pair of
consecutive hops, per model.  It also counts how a greedy pairing (pair when allowed, else single) shortens the chain.

    python tools/exp/double_hop_rate.py c5 4 2      # workload, batches per launch, launches
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import pyoracle  # noqa: E402
from zebra_amd import synth  # noqa: E402


def side(row, j, scale, k):
    n = int(row["len"][j]) if row["norm"][j] != 0.0 else 0
    keys = list(zip(row["eidx"][j][:n].tolist(), row["node"][j][:n].tolist(), row["ts"][j][:n].tolist()))
    return keys, row["w"][j][:n] * scale


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "c5"
    group = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    launches = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    wl = synth.WORKLOADS[name]
    bs, k = wl["bs"], wl["k"]
    prefill = (wl["n_edges"] // 10) // bs
    n_edges = (prefill + group * launches) * bs
    src, dst, ts, eidx = synth.power_law_stream(wl["n_nodes"], n_edges, bipartite=wl["bipartite"], seed=2020, perm_seed=7)
    M = len(wl["alpha"])
    o = pyoracle.TpprOracle(wl["n_nodes"] + 1, k, M, wl["alpha"], wl["beta"])
    t0 = time.time()
    step = 64 * bs
    for a in range(0, prefill * bs, step):
        b = min(prefill * bs, a + step)
        o.update_only(src[a:b], dst[a:b], ts[a:b], eidx[a:b])
    print("prefill %d batches in %.0f s" % (prefill, time.time() - t0), flush=True)
    tot = {m: dict(hops=0, pairs=0, a=0, b=0, c=0, d=0, ok=0, sections=0) for m in range(M)}
    for L in range(launches):
        a0 = (prefill + L * group) * bs
        a1 = a0 + group * bs
        ends = np.concatenate([src[a0:a1], dst[a0:a1]])
        ids, cnt = np.unique(ends, return_counts=True)
        hub = int(ids[np.argmax(cnt)])
        pos = [i for i in range(a0, a1) if src[i] == hub or dst[i] == hub]
        recs = {m: [] for m in range(M)}
        cur = a0
        for i in pos:
            if i > cur:
                o.update_only(src[cur:i], dst[cur:i], ts[cur:i], eidx[cur:i])
            p = int(dst[i]) if src[i] == hub else int(src[i])
            for m in range(M):
                r = o.export_rows(m, [hub, p])
                recs[m].append((i, p, r))
            o.update_only(src[i:i + 1], dst[i:i + 1], ts[i:i + 1], eidx[i:i + 1])
            cur = i + 1
        if cur < a1:
            o.update_only(src[cur:a1], dst[cur:a1], ts[cur:a1], eidx[cur:a1])
        for m in range(M):
            alpha, beta = wl["alpha"][m], wl["beta"][m]
            R = recs[m]
            T = tot[m]
            T["hops"] += len(R)
            allowed = []
            for q in range(len(R) - 1):
                (i1, p1, r1), (i2, p2, r2) = R[q], R[q + 1]
                T["pairs"] += 1
                hn = r1["norm"][0]
                lean1 = p1 != hub and hn != 0.0 and r1["len"][0] == k
                lean2 = p2 != hub and r2["len"][0] == k
                okA = lean1 and lean2
                okB = p1 != p2
                if not okA:
                    allowed.append(False)
                    continue
                T["a"] += 1
                nn = hn * beta + beta
                s1, s2 = hn / nn * beta, beta / nn * (1 - alpha)
                nn2 = nn * beta + beta
                s1b, s2b = nn / nn2 * beta, beta / nn2 * (1 - alpha)
                hk, hw = side(r1, 0, s1, k)
                k1, w1 = side(r1, 1, s2, k)
                k2, w2 = side(r2, 1, s2b, k)        # p2's row as hop t+1 finds it (p1 != p2: not touched by hop t)
                new1 = (int(eidx[i1]), p1, float(ts[i1]))
                new2 = (int(eidx[i2]), p2, float(ts[i2]))
                S0, S1, S2 = set(hk), set(k1) | {new1}, set(k2) | {new2}
                okC = not (S0 & S1) and not (S0 & S2) and not (S1 & S2)
                # final weights of the 62 candidates
                fw = np.concatenate([hw * s1b, w1 * s1b, [(s2 * alpha if alpha != 0 else s2) * s1b], w2,
                                     [s2b * alpha if alpha != 0 else s2b]])
                fs = np.sort(fw)
                drop = len(fs) - k
                okD = drop > 0 and fs[drop] != fs[drop - 1]
                T["b"] += okB
                T["c"] += okB and okC
                T["d"] += okD
                ok = okB and okC and okD
                T["ok"] += ok
                allowed.append(bool(ok))
            # greedy pairing along the chain
            q, sections = 0, 0
            while q < len(R):
                if q < len(allowed) and allowed[q]:
                    q += 2
                else:
                    q += 1
                sections += 1
            T["sections"] += sections
        print("launch %d: hub %d, %d hops" % (L, hub, len(pos)), flush=True)
    for m in range(M):
        T = tot[m]
        print("model %d (beta %.2f): hops %d  consecutive pairs %d  (a) both lean-shaped %d  (b) + partners differ %d  "
              "(c) + keys disjoint %d  (d) final cut not in a run %d  ALL %d = %.1f %%  greedy pairing: %d sections for %d hops = %.2f"
              % (m, wl["beta"][m], T["hops"], T["pairs"], T["a"], T["b"], T["c"], T["d"], T["ok"],
                 100.0 * T["ok"] / max(1, T["pairs"]), T["sections"], T["hops"], T["sections"] / max(1, T["hops"])))


if __name__ == "__main__":
    main()
