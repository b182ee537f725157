# Why does a hub hop wait for its partner's row?  (-DZT_STAMP build, tools/build_stamp.sh.)  For every hop of the busiest
# hub in the last launch whose preparation (entered -> partner's row in hand) is long, find the task that wrote that row
# -- the partner's previous edge e' in the launch -- and say when IT was dequeued, had its inputs and stored, all on the
# hop's own time axis (100 MHz ticks -> us).  T-PPR alone, warm state, model 0.
#   python tools/exp/partner_waits.py [launches] [workload] [batches per launch] [slow threshold us]
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
SLOW = float(sys.argv[4]) if len(sys.argv) > 4 else 10.0
wl = synth.WORKLOADS[WL]; B = wl["bs"] * G
src, dst, ts, eidx = synth.power_law_stream(wl["n_nodes"], NB * B, bipartite=wl["bipartite"], seed=2020, perm_seed=7)
neg = synth.negatives(dst, len(src), seed=2021)
f = tppr.tppr_finder(wl["n_nodes"] + 1, 20, 2, [0.1, 0.1], [0.5, 0.95])
d = torch.device('cuda')
sd, dd, nd = [torch.from_numpy(x).to(d) for x in (src, dst, neg)]
td, ed = torch.from_numpy(ts).to(d), torch.from_numpy(eidx).to(d)


def launch(b):
    s, e = b * B, (b + 1) * B
    f.stream_device(torch.cat([sd[s:e], dd[s:e], nd[s:e]]), td[s:e], ed[s:e], 3, True, -1, check_status=False)


def stamps():
    a = np.zeros((B, 4), np.int64); c = np.zeros((B, 12), np.int64)
    lib.zt_debug_stamps(a.ctypes.data_as(C.c_void_p), C.c_int(B))
    lib.zt_debug_stamps3(c.ctypes.data_as(C.c_void_p), C.c_int(B))
    return a, c


for b in range(NB - 1):
    launch(b)
f.check_status()
a0, c0 = stamps()
before = max(a0.max(), c0.max())                  # anything later belongs to the last launch
launch(NB - 1)
f.check_status()
ge, ch = stamps()
ge = np.where(ge > before, ge, 0); ch = np.where(ch > before, ch, 0)
t00 = min(ge[ge > 0].min(), ch[ch > 0].min())
us = lambda x: (x - t00) * 0.01
s0 = (NB - 1) * B
u, v = src[s0:s0 + B], dst[s0:s0 + B]
cnt = np.bincount(np.concatenate([u, v]))
hub = cnt.argmax()
is_hop = ch[:, 0] > 0; is_half = ch[:, 4] > 0; is_gen = ge[:, 0] > 0
print("%s, %d edges per launch: %d hub hops stamped (all chains), %d partner halves, %d tasks through process_edge; launch span %.1f us" % (
    WL, B, is_hop.sum(), is_half.sum(), is_gen.sum(), us(max(ge.max(), ch.max()))))
idx = np.where(((u == hub) | (v == hub)) & is_hop)[0]
part = np.where(u[idx] == hub, v[idx], u[idx])
prep = (ch[idx, 1] - ch[idx, 0]) * 0.01
turn = (ch[idx, 2] - ch[idx, 1]) * 0.01
p = lambda x: np.percentile(x, [10, 50, 90, 99]).round(1)
print("busiest hub %d: %d hops; preparation (entered -> partner's row in hand) us 10/50/90/99 %%: %s ; then -> turn arrived %s" % (hub, len(idx), p(prep), p(turn)))
gen_deq = np.where(is_gen, ge[:, 0], 0)
order = np.argsort(np.where(gen_deq > 0, gen_deq, np.iinfo(np.int64).max))
deq_sorted = gen_deq[order][: int((gen_deq > 0).sum())]
edge_sorted = order[: len(deq_sorted)]
run_max = np.maximum.accumulate(edge_sorted)


def queue_head(t):                                # the largest edge index dequeued through process_edge by time t
    j = np.searchsorted(deq_sorted, t, side='right') - 1
    return int(run_max[j]) if j >= 0 else -1


slow = np.where(prep > SLOW)[0]
print("hops with preparation > %.0f us: %d, sum %.0f us of the chain's %.0f us" % (SLOW, len(slow), prep[slow].sum(), us(ch[idx, 3].max()) - us(ch[idx, 0].min())))
kinds = {}
for t in slow:
    e = idx[t]; pn = part[t]
    w = np.where((u[:e] == pn) | (v[:e] == pn))[0]
    h0, h1 = ch[e, 0], ch[e, 1]
    head = queue_head(h0)
    if len(w) == 0:
        kind = "none"; line = "no writer in this launch"
    else:
        w = int(w[-1])
        if is_half[w]:
            other = v[w] if u[w] == pn else u[w]
            r = ch[w, 4:8]
            late = r[0] > h0
            kind = "half:" + ("dequeued after the hop entered" if late else ("waited for its hub's version" if (r[2] - r[1]) > (r[1] - r[0]) else "waited for its own rows"))
            line = "partner half of edge %d (hub %d, %d edges back): entered %+.1f, rows %+.1f, version %+.1f, stored %+.1f" % (
                w, other, e - w, *[(x - h0) * 0.01 for x in r])
        elif is_gen[w]:
            r = ge[w]
            late = r[0] > h0
            kind = "general:" + ("dequeued after the hop entered" if late else "dequeued before, waited for its rows")
            line = "general task of edge %d (%d edges back): dequeued %+.1f, rows ready %+.1f, first row stored %+.1f, end %+.1f" % (
                w, e - w, *[(x - h0) * 0.01 for x in r])
        elif is_hop[w]:
            kind = "hop"; line = "a hub hop of edge %d (partner is a hub itself)" % w
        else:
            kind = "unstamped"; line = "edge %d not stamped" % w
    kinds[kind] = kinds.get(kind, 0) + 1
    print("  t=%3d edge %4d entered at %7.1f us, row in hand +%.1f; queue head then: edge %d (%+d); writer: %s" % (t, e, us(h0), (h1 - h0) * 0.01, head, head - e, line))
print("causes: " + "; ".join("%s %d" % kv for kv in sorted(kinds.items())))
# how far is the in-order queue behind / ahead of the chain over the launch?
q = max(1, len(idx) // 8)
print("queue head minus the hub's edge at hop entry, every %d hops: " % q + " ".join("%+d" % (queue_head(ch[idx[t], 0]) - idx[t]) for t in range(0, len(idx), q)))

# ---- every chain: hops, span, and what its hops waited for ----
# hop t's turn can come only after publication(t-1): the time between that and its own critical section's start is time
# the chain stood still for THIS hop (its partner's row, or the wave was late); publication(t) - start(t) is the section.
nodes = np.unique(np.concatenate([u[is_hop], v[is_hop]]))
rows = []
for hnode in nodes:
    ii = np.where(((u == hnode) | (v == hnode)) & is_hop)[0]
    if len(ii) < 8:
        continue
    own = ii[ch[ii, 8] > 0]
    fb = int((((u == hnode) | (v == hnode)) & is_gen & ~is_hop).sum())
    st, pb, rw = ch[own, 2], ch[own, 8], ch[own, 1]
    stand = (st[1:] - pb[:-1]) * 0.01
    rowlate = ((rw[1:] - pb[:-1]) * 0.01).clip(min=0)
    sec = (pb - st) * 0.01
    rows.append((len(ii), hnode, len(own), fb, us(ch[ii, 0].min()), us(ch[ii, 8].max()), np.median(sec), sec.sum(), stand.clip(min=0).sum(), rowlate.sum(), int((rowlate > 5).sum())))
print("chains (by hops): hub, hops, published lean/split, edges through process_edge, first entered us, last publication us, median section us, sum of sections, sum of standstill between sections, of which the partner's row came after the predecessor's publication (sum, hops > 5 us)")
for r in sorted(rows, reverse=True):
    print("  hub %6d hops %4d pub %4d fallback %3d  %7.1f -> %7.1f us  section med %.2f sum %6.1f  standstill %6.1f  row-late %6.1f (%d hops)" % (r[1], r[0], r[2], r[3], r[4], r[5], r[6], r[7], r[8], r[9], r[10]))
# the hub's version as the partner halves see it: publication of hop t-1 -> "version in hand" of the half of hop t
lag = []
for hnode in nodes:
    ii = np.where(((u == hnode) | (v == hnode)) & is_hop & is_half)[0]
    for a, b in zip(ii[:-1], ii[1:]):
        if ch[a, 8] > 0 and ch[b, 6] > 0 and ch[b, 5] < ch[a, 8]:          # the half was already waiting
            lag.append((ch[b, 6] - ch[a, 8]) * 0.01)
if lag:
    print("publication of a hop -> its successor's partner half has the version (halves already waiting): us 10/50/90/99 %%: %s  (n=%d)" % (p(np.asarray(lag)), len(lag)))
m2 = is_half & (ch[:, 7] > 0)
print("partner half: version in hand -> partner's row stored us %s ; entered -> rows %s" % (p((ch[m2, 7] - ch[m2, 6]) * 0.01), p((ch[m2, 5] - ch[m2, 4]) * 0.01)))

# ---- the measured critical path, walked back from the last publication ----
cnt3 = np.bincount(np.concatenate([u, v, neg[s0:s0 + B]]), minlength=wl["n_nodes"] + 2)
hubset = [int(x) for x in np.argsort(-cnt3, kind="stable")[:16] if cnt3[x] >= 24]
owner = np.full(B, -1)
for e in range(B):
    a, b = int(u[e]), int(v[e])
    ia, ib = a in hubset, b in hubset
    if ia and (not ib or cnt3[a] >= cnt3[b]): owner[e] = a
    elif ib: owner[e] = b
prev_in_chain = np.full(B, -1)
last = {}
for e in range(B):
    if owner[e] >= 0:
        prev_in_chain[e] = last.get(owner[e], -1); last[owner[e]] = e
prev_writer = np.full((B, 3), -1)                        # previous writer of u, v and of the negative's row
lastw = {}
ng = neg[s0:s0 + B]
for e in range(B):
    for r, x in enumerate((int(u[e]), int(v[e]), int(ng[e]))):
        prev_writer[e, r] = lastw.get(x, -1)
    lastw[int(u[e])] = e; lastw[int(v[e])] = e


def stored_at(w, node):
    """when edge w's new row of `node` was stored (ticks), and by what"""
    if owner[w] >= 0 and is_hop[w] and ch[w, 8] > 0:
        if node == owner[w]:
            return ch[w, 3], "hop"                         # the hub's row to memory: end of the hop (if it goes there at all)
        return (ch[w, 7] if ch[w, 7] > 0 else ch[w, 6]), "half"
    return ge[w, 2] if ge[w, 2] > 0 else ge[w, 3], "gen"


acc = {}
def add(k, ticks):
    acc[k] = acc.get(k, 0.0) + ticks * 0.01

pubs = np.where(ch[:, 8] > 0, ch[:, 8], 0)
e = int(pubs.argmax()); kind = "hop"; t_now = pubs[e]
steps = 0; trail = []
VERBOSE = len(sys.argv) > 5
while e >= 0 and steps < 5000:
    steps += 1
    if kind == "hop":
        st, rowt, ent = ch[e, 2], ch[e, 1], ch[e, 0]
        add("hub hop: critical section", t_now - st)
        pe = prev_in_chain[e]
        ppub = ch[pe, 8] if pe >= 0 and ch[pe, 8] > 0 else (ge[pe, 3] if pe >= 0 else 0)
        if rowt > ppub:                                   # the partner's row (and its preparation) came last
            add("hub hop: prepared -> turn", st - rowt)
            pn = int(v[e]) if int(u[e]) == owner[e] else int(u[e])
            w = prev_writer[e, 1 if int(u[e]) == owner[e] else 0]
            sw, kw = stored_at(w, pn) if w >= 0 else (0, "none")
            if VERBOSE: print("    hop %d (hub %d, partner %d): entered %.1f prepared %.1f turn %.1f pub %.1f; writer %d (%s) stored %.1f" % (e, owner[e], pn, us(ent), us(rowt), us(st), us(ch[e, 8]), w, kw, us(sw)))
            if sw < ent:                                  # the row was there; the hop was entered late (wave busy)
                add("hub hop: entered -> prepared", rowt - ent); t_now = ent; kind = "hop-entry"
                continue
            add("row stored -> consumer hop prepared", rowt - sw); t_now = sw; e = w; kind = kw; trail.append(kw)
        else:
            add("hub hop: publication -> successor's turn", st - ppub)
            if pe < 0: break
            t_now = ppub; e = pe; kind = "hop" if ch[pe, 8] > 0 else "gen"
    elif kind == "hop-entry":
        # a wave entered hop e late: it had been busy with a hop eight positions back; follow the chain's publication
        # the wave that claimed hop e had just finished one of the previous eight hops of the chain
        q = prev_in_chain[e]; best = -1
        for _ in range(8):
            if q < 0: break
            endq = ch[q, 3] if ch[q, 3] > 0 else ge[q, 3]
            if endq <= t_now and (best < 0 or endq > bend): best, bend = q, endq
            q = prev_in_chain[q]
        if best < 0: trail.append("start"); break
        add("hub hop: a wave free -> next hop entered", t_now - bend)
        if ch[best, 8] > 0:
            add("hub hop: publication -> end (tail)", bend - ch[best, 8]); t_now = ch[best, 8]; e = best; kind = "hop"
        else:
            t_now = bend; e = best; kind = "gen"
    elif kind == "half":
        ent, rows_t, ver, sto = ch[e, 4], ch[e, 5], ch[e, 6], ch[e, 7]
        add("partner half: version in hand -> row stored", t_now - ver)
        pe = prev_in_chain[e]
        ppub = ch[pe, 8] if pe >= 0 and ch[pe, 8] > 0 else (ge[pe, 3] if pe >= 0 else 0)
        if ver - rows_t > 50 or pe < 0 and False:         # it stood waiting for the version
            add("publication -> partner half has the version", ver - ppub)
            if pe < 0: break
            t_now = ppub; e = pe; kind = "hop" if ch[pe, 8] > 0 else "gen"
        else:                                             # its own rows (or its dequeue) came last
            pn_role = 1 if int(u[e]) == owner[e] else 0
            cands = [(stored_at(w, x) + (w,)) for w, x in ((prev_writer[e, pn_role], int(v[e]) if pn_role else int(u[e])), (prev_writer[e, 2], int(ng[e]))) if w >= 0]
            cands = [c for c in cands if c[0] > ent]
            if not cands:
                add("partner half: dequeued late", ver - ent); trail.append("queue"); break
            sw, kw, w = max(cands)
            add("row stored -> partner half has its rows", ver - sw); t_now = sw; e = w; kind = kw
    else:                                                 # a task through process_edge
        deq, rows_t, sto = ge[e, 0], ge[e, 1], ge[e, 2] if ge[e, 2] > 0 else ge[e, 3]
        add("general task: rows ready -> first row stored", t_now - rows_t)
        cands = [(stored_at(w, x) + (w,)) for w, x in zip(prev_writer[e], (int(u[e]), int(v[e]), int(ng[e]))) if w >= 0]
        cands = [c for c in cands if c[0] > deq]
        if not cands:
            add("general task: dequeued -> rows ready (no writer after its dequeue)", rows_t - deq); trail.append("queue"); break
        sw, kw, w = max(cands)
        add("row stored -> general task has its rows", rows_t - sw); t_now = sw; e = w; kind = kw
print("critical path walked back from the last publication (%.1f us): %d steps, ends at %s" % (us(pubs.max()), steps, trail[-1] if trail else "?"))
for k, x in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("   %-62s %7.1f us" % (k, x))
print("   sum %.1f us" % sum(acc.values()))
