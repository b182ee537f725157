# What do the hub chains of a launch wait for?  (-DZT_STAMP build, tools/build_stamp.sh.)  Stamps (100 MHz ticks, model 0)
# of every hub hop by (chain, position), of every partner half and every general task by edge; from them: per chain the
# time its hops stood still and why, the hand-off costs between chains, and the measured critical path walked back from
# the last publication.  T-PPR alone, warm state.
#   python tools/exp/partner_waits.py [launches] [workload] [batches per launch] [v]
import sys, ctypes as C
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from zebra_amd import _capi
_capi.LIB_PATH = '/root/repo/tools/out/libzebra_stamp.so'
from zebra_amd import tppr, synth
lib = _capi.lib()
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 100
WL = sys.argv[2] if len(sys.argv) > 2 else "c3"
G = int(sys.argv[3]) if len(sys.argv) > 3 else 4
VERBOSE = len(sys.argv) > 4
wl = synth.WORKLOADS[WL]; B = wl["bs"] * G
src, dst, ts, eidx = synth.power_law_stream(wl["n_nodes"], NB * B, bipartite=wl["bipartite"], seed=2020, perm_seed=7)
neg = synth.negatives(dst, len(src), seed=2021)
f = tppr.tppr_finder(wl["n_nodes"] + 1, 20, 2, [0.1, 0.1], [0.5, 0.95])
d = torch.device('cuda')
sd, dd, nd = [torch.from_numpy(x).to(d) for x in (src, dst, neg)]
td, ed = torch.from_numpy(ts).to(d), torch.from_numpy(eidx).to(d)
CH, CHM = 16, 2048


def launch(b):
    s, e = b * B, (b + 1) * B
    f.stream_device(torch.cat([sd[s:e], dd[s:e], nd[s:e]]), td[s:e], ed[s:e], 3, True, -1, check_status=False)


def stamps():
    a = np.zeros((B, 4), np.int64); c = np.zeros((B, 8), np.int64); hp = np.zeros((CH, CHM, 8), np.int64)
    lib.zt_debug_stamps(a.ctypes.data_as(C.c_void_p), C.c_int(B))
    lib.zt_debug_stamps3(c.ctypes.data_as(C.c_void_p), C.c_int(B))
    lib.zt_debug_hopst(hp.ctypes.data_as(C.c_void_p))
    return a, c, hp


for b in range(NB - 1):
    launch(b)
f.check_status()
a0, c0, h0 = stamps()
before = max(a0.max(), c0.max(), h0.max())        # anything later belongs to the last launch
launch(NB - 1)
f.check_status()
ge, ph, hp = stamps()
ge = np.where(ge > before, ge, 0); ph = np.where(ph > before, ph, 0); hp = np.where(hp > before, hp, 0)
t00 = min(x[x > 0].min() for x in (ge, ph, hp))
us = lambda x: (x - t00) * 0.01
s0 = (NB - 1) * B
u, v, g = src[s0:s0 + B], dst[s0:s0 + B], neg[s0:s0 + B]
# ---- the plan, restated (tppr_prepass.hip): chains, positions, owners ----
cnt = {}
for i in range(B):
    for x in {int(u[i]), int(v[i]), int(g[i])}:
        cnt[x] = cnt.get(x, 0) + 1
hot = sorted([x for x in cnt if cnt[x] >= 24], key=lambda x: (-cnt[x], x))[:CH]
chain_of = {x: c for c, x in enumerate(hot)}
wr = {}
pos = np.full((B, 2), -1)                          # writer ordinal of u / v at each edge
for i in range(B):
    pos[i, 0] = wr.get(int(u[i]), 0); pos[i, 1] = wr.get(int(v[i]), 0)
    for x in {int(u[i]), int(v[i])}:
        wr[x] = wr.get(x, 0) + 1
edges_of = [[i for i in range(B) if u[i] == x or v[i] == x] for x in hot]
owner = np.full(B, -1)
for i in range(B):
    a, b = int(u[i]), int(v[i])
    ia, ib = a in chain_of, (b in chain_of and b != a)
    if ia and (not ib or cnt[a] >= cnt[b]): owner[i] = chain_of[a]
    elif ib: owner[i] = chain_of[b]
last_writer = {}
prev_w = np.full((B, 3), -1)                       # previous edge that wrote u's / v's / the negative's row
for i in range(B):
    for r, x in enumerate((int(u[i]), int(v[i]), int(g[i]))):
        prev_w[i, r] = last_writer.get(x, -1)
    last_writer[int(u[i])] = i; last_writer[int(v[i])] = i
n_dual = sum(1 for i in range(B) if int(u[i]) in chain_of and int(v[i]) in chain_of and u[i] != v[i])
span = us(max(ge.max(), ph.max(), hp.max()))
print("%s, %d edges per launch: %d chains, %d edges with a hub endpoint, %d between two hubs; launch span %.1f us" % (
    WL, B, len(hot), int((owner >= 0).sum()), n_dual, span))
p = lambda x: np.percentile(x, [10, 50, 90, 99]).round(1) if len(x) else "-"
# ---- per chain ----
print("chains: hub, hops, first entered -> last publication (us), sum of critical sections, standstill between sections, of it: the partner's row came after the predecessor's publication (sum, hops > 5 us)")
for c, x in enumerate(hot):
    n = len(edges_of[c]); h = hp[c, :n]
    ok = (h[:, 4] > 0) & (h[:, 2] > 0)
    if ok.sum() < 2: continue
    st, pb, rw = h[:, 2], h[:, 4], h[:, 1]
    sec = ((pb - st) * 0.01)[ok]
    okk = ok[1:] & ok[:-1]
    stand = ((st[1:] - pb[:-1]) * 0.01)[okk].clip(min=0)
    late = ((rw[1:] - pb[:-1]) * 0.01)[okk].clip(min=0)
    print("  hub %8d hops %4d  %7.1f -> %7.1f  sections %6.1f (median %.2f)  standstill %6.1f  row-late %6.1f (%d hops)" % (
        x, n, us(h[ok, 0].min()), us(pb[ok].max()), sec.sum(), np.median(sec), stand.sum(), late.sum(), int((late > 5).sum())))
# ---- hand-off costs ----
lag_ver, lag_hub = [], []
for c, x in enumerate(hot):
    E = edges_of[c]
    for t in range(1, len(E)):
        e = E[t]
        if hp[c, t - 1, 4] == 0: continue
        if owner[e] == c and ph[e, 6] > 0 and ph[e, 5] < hp[c, t - 1, 4]:            # the half was already waiting
            lag_ver.append((ph[e, 6] - hp[c, t - 1, 4]) * 0.01)
        # the same version as another hub's chain sees it: that chain's hop of edge e (a position of both chains)
        a, b = int(u[e]), int(v[e])
        o = b if a == x else a
        if o in chain_of and o != x:
            c2 = chain_of[o]; t2 = pos[e, 1] if a == x else pos[e, 0]
            if hp[c2, t2, 1] > 0 and hp[c2, t2, 0] < hp[c, t - 1, 4]:
                lag_hub.append((hp[c2, t2, 1] - hp[c, t - 1, 4]) * 0.01)
print("publication of a hop -> the next edge's partner half has the version (halves already waiting) us 10/50/90/99 %%: %s (n=%d)" % (p(np.asarray(lag_ver)), len(lag_ver)))
print("publication of a hop -> another hub's hop of the next edge is prepared (it was already waiting): %s (n=%d)" % (p(np.asarray(lag_hub)), len(lag_hub)))
tails = np.concatenate([((hp[c, :len(edges_of[c]), 3] - hp[c, :len(edges_of[c]), 4]) * 0.01)[(hp[c, :len(edges_of[c]), 4] > 0) & (hp[c, :len(edges_of[c]), 3] > 0)] for c in range(len(hot))])
preps = np.concatenate([((hp[c, :len(edges_of[c]), 1] - hp[c, :len(edges_of[c]), 0]) * 0.01)[(hp[c, :len(edges_of[c]), 1] > 0) & (hp[c, :len(edges_of[c]), 0] > 0)] for c in range(len(hot))])
print("hub hop: publication -> end (replay, order, version stored) %s ; entered -> prepared %s" % (p(tails), p(preps)))
m2 = ph[:, 7] > 0
print("partner half: version in hand -> partner's row stored %s" % p((ph[m2, 7] - ph[m2, 6]) * 0.01))
# ---- the measured critical path, walked back from the last publication ----
acc = {}
def add(k, ticks): acc[k] = acc.get(k, 0.0) + ticks * 0.01


def row_event(e, r):
    """who made the row that edge e's role r (0 u, 1 v, 2 negative) reads, and when it was stored"""
    x = int((u, v, g)[r][e])
    if x in chain_of:                                  # by version: the chain's hop before that position
        c2 = chain_of[x]
        t2 = pos[e, r] if r < 2 else sum(1 for q in edges_of[c2] if q < e)
        if t2 == 0: return None
        return ("hop", c2, t2 - 1, hp[c2, t2 - 1, 3] if hp[c2, t2 - 1, 3] > 0 else hp[c2, t2 - 1, 4])
    w = prev_w[e, r]
    if w < 0: return None
    if owner[w] >= 0: return ("half", w, 0, ph[w, 7] if ph[w, 7] > 0 else ph[w, 6])
    return ("gen", w, 0, ge[w, 2] if ge[w, 2] > 0 else ge[w, 3])


def follow(ev):
    global kind, a1, a2, t_now
    if ev[0] == "hop":
        add("hub hop: publication -> version stored (tail)", ev[3] - hp[ev[1], ev[2], 4])
        kind, a1, a2, t_now = "hop", ev[1], ev[2], hp[ev[1], ev[2], 4]
    else:
        kind, a1, t_now = ev[0], ev[1], ev[3]


cbest = max(range(len(hot)), key=lambda c: hp[c, :, 4].max())
tbest = int(hp[cbest, :, 4].argmax())
kind, a1, a2 = "hop", cbest, tbest
t_now = hp[cbest, tbest, 4]
steps = 0
while steps < 20000:
    steps += 1
    if kind == "hop":
        c, t = a1, a2
        e = edges_of[c][t]; h = hp[c, t]
        ent, prep, st = h[0], h[1], h[2]
        add("hub hop: critical section", t_now - st)
        ppub = hp[c, t - 1, 4] if t > 0 else 0
        if t > 0 and prep <= ppub:
            add("hub hop: publication -> successor's section starts", st - ppub)
            t_now = ppub; a2 = t - 1
            continue
        add("hub hop: prepared -> section starts", st - prep)
        r = 1 if int(u[e]) == hot[c] else 0
        ev = row_event(e, r)
        if VERBOSE: print("    hop chain %d pos %d edge %d: entered %.1f prepared %.1f turn %.1f pub %.1f; partner's row: %s" % (c, t, e, us(ent), us(prep), us(st), us(h[4]), ev and (ev[0], ev[1], ev[2], round(us(ev[3]), 1))))
        if ev is None or ev[3] < ent:                  # the row was there: the hop was entered late (its wave was busy)
            add("hub hop: entered -> prepared", prep - ent)
            best = None
            for q in range(max(0, t - 8), t):
                if 0 < hp[c, q, 3] <= ent and (best is None or hp[c, q, 3] > hp[c, best, 3]): best = q
            if best is None: print("   (walk ends: first hops of chain %d)" % c); break
            add("hub hop: a wave free -> its next hop entered", ent - hp[c, best, 3])
            add("hub hop: publication -> end (tail)", hp[c, best, 3] - hp[c, best, 4])
            t_now = hp[c, best, 4]; a2 = best
            continue
        add("row / version stored -> consumer hop prepared", prep - ev[3])
        follow(ev)
    elif kind == "half":
        e = a1; c = owner[e]
        t = pos[e, 0] if int(u[e]) == hot[c] else pos[e, 1]
        ent, rows_t, ver = ph[e, 4], ph[e, 5], ph[e, 6]
        add("partner half: version in hand -> row stored", t_now - ver)
        if t > 0 and ver - rows_t > 50:
            add("publication -> partner half has the version", ver - hp[c, t - 1, 4])
            kind, a1, a2, t_now = "hop", c, t - 1, hp[c, t - 1, 4]
            continue
        r = 1 if int(u[e]) == hot[c] else 0
        cands = [x for x in (row_event(e, r), row_event(e, 2)) if x is not None and x[3] > ent]
        if not cands:
            add("partner half: dequeued -> rows", ver - ent); print("   (walk ends: a partner half dequeued late, edge %d at %.1f us)" % (e, us(ent))); break
        ev = max(cands, key=lambda x: x[3])
        add("row / version stored -> partner half has its rows", ver - ev[3])
        follow(ev)
    else:
        e = a1
        deq, rows_t = ge[e, 0], ge[e, 1]
        add("general task: rows ready -> first row stored", t_now - rows_t)
        cands = [x for x in (row_event(e, 0), row_event(e, 1), row_event(e, 2)) if x is not None and x[3] > deq]
        if not cands:
            add("general task: dequeued -> rows ready", rows_t - deq); print("   (walk ends: a general task dequeued late, edge %d at %.1f us)" % (e, us(deq))); break
        ev = max(cands, key=lambda x: x[3])
        add("row / version stored -> general task has its rows", rows_t - ev[3])
        follow(ev)
print("critical path walked back from the last publication (%.1f us, chain of hub %d): %d steps" % (us(hp[cbest, tbest, 4]), hot[cbest], steps))
for k, x in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("   %-66s %7.1f us" % (k, x))
print("   sum %.1f us" % sum(acc.values()))
