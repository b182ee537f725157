# The dependency graph of one launch of k_stream under a cost model (CPU only): which costs make the launch's span?
# Chains = the 16 most-touched nodes with >= 24 accesses.  A chain-owned edge has a hub hop (critical section CS after the
# later of the predecessor hop's publication and the partner's row + PREP) and a partner half (MERGE after the later of
# the hub's version = publication of the previous hop + VER, and the partner's old row + VIS); a general edge stores GEN
# after the later of its two rows + VIS.  The in-order queue is taken as never behind (it is ~1000 edges ahead on C3).
#   python tools/exp/chain_dag.py [workload] [batches per launch] [launch index]
import sys
sys.path.insert(0, '/root/repo')
import numpy as np
from zebra_amd import synth
WL = sys.argv[1] if len(sys.argv) > 1 else "c3"
G = int(sys.argv[2]) if len(sys.argv) > 2 else 4
LI = int(sys.argv[3]) if len(sys.argv) > 3 else 99
wl = synth.WORKLOADS[WL]; B = wl["bs"] * G
src, dst, ts, eidx = synth.power_law_stream(wl["n_nodes"], (LI + 1) * B, bipartite=wl["bipartite"], seed=2020, perm_seed=7)
u, v = src[LI * B:], dst[LI * B:]
cnt = np.bincount(np.concatenate([u, v]))
cand = np.argsort(-cnt, kind="stable")[:16]
hubs = set(int(x) for x in cand if cnt[x] >= 24)


def span(CS=1.0, PREP=3.5, VER=2.0, VIS=1.5, MERGE=4.0, GEN=5.0, HAND=0.1, trace=False):
    row = {}                       # node -> (time its row of now is readable, what wrote it)
    pub = {}                       # hub -> publication time of its latest hop
    why = [None] * B
    end = 0.0
    for e in range(B):
        a, b = int(u[e]), int(v[e])
        h = a if a in hubs else (b if b in hubs else None)
        if h is None:
            ra, rb = row.get(a, (0.0, -1)), row.get(b, (0.0, -1))
            t = max(ra[0], rb[0]) + VIS + GEN
            row[a] = (t, e); row[b] = (t, e)
            why[e] = ("gen", ra[1] if ra[0] >= rb[0] else rb[1])
        else:
            p = b if h == a else a
            rp = row.get(p, (0.0, -1))
            prev = pub.get(h, (0.0, -1))
            t_row = rp[0] + PREP
            t_prev = prev[0] + HAND
            start = max(t_row, t_prev)
            t_pub = start + CS
            # the partner's new row: from the hub's OLD row (version = the previous hop's publication + VER)
            if p != h and p not in hubs:
                t_half = max(prev[0] + VER, rp[0] + VIS) + MERGE
                row[p] = (t_half, e)
            elif p in hubs and p != h:
                # hub-hub edge: the other hub's row goes through its own chain too; approximate as one more hop there
                pp = pub.get(p, (0.0, -1))
                t2 = max(pp[0], t_pub) + CS
                pub[p] = (t2, e)
            pub[h] = (t_pub, e)
            why[e] = ("hop-row", rp[1]) if t_row > t_prev else ("hop-chain", prev[1])
            t = t_pub
        end = max(end, t)
    if trace:
        # walk the critical path back from the last hop of the slowest chain
        e = max(pub.values())[1]
        kinds = {}
        while e is not None and e >= 0 and why[e] is not None:
            k, nxt = why[e]
            kinds[k] = kinds.get(k, 0) + 1
            e = nxt
        return end, kinds
    return end


base, kinds = span(trace=True)
print("%s launch %d, %d edges, %d chains (busiest %d hops), chain-owned edges %d" % (WL, LI, B, len(hubs), cnt.max(), sum(1 for e in range(B) if int(u[e]) in hubs or int(v[e]) in hubs)))
print("model span %.0f us; steps on the critical path: %s" % (base, kinds))
for name, kw in [("critical section 0.5", dict(CS=0.5)), ("preparation 1.75", dict(PREP=1.75)), ("version 1.0", dict(VER=1.0)), ("visibility 0.75", dict(VIS=0.75)),
                 ("merge 2.0", dict(MERGE=2.0)), ("general 2.5", dict(GEN=2.5)), ("all cross-chain costs halved", dict(PREP=1.75, VER=1.0, VIS=0.75, MERGE=2.0)),
                 ("no cross-chain cost at all", dict(PREP=0, VER=0, VIS=0, MERGE=0, GEN=0)), ("critical section only, 1.0", dict(PREP=0, VER=0, VIS=0, MERGE=0, GEN=0, HAND=0))]:
    print("  %-36s -> %.0f us" % (name, span(**kw)))
