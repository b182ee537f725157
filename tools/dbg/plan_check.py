# debug: dump the plan of one launch and check it against a numpy restatement
import sys, ctypes as C
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/tests/golden')
import numpy as np, torch
import inputs as I
from zebra_amd import _capi
import os
if os.environ.get('ZT_DBG_LIB'): _capi.LIB_PATH = os.environ['ZT_DBG_LIB']
from zebra_amd import tppr
from zebra_amd._capi import ptr
name = sys.argv[1] if len(sys.argv) > 1 else "bip_k20"
kind, N, E, seed, bs, k, al, be, _ = I.STREAM_CASES[name]
src, dst, neg, ts, eidx = I.make_stream(kind, N, E, seed)
f = tppr.tppr_finder(N, k, len(al), al, be)
DBG = bool(os.environ.get('ZT_DBG_LIB'))
hooks = None if DBG else _capi.hooks_lib()
for b in range((E + bs - 1) // bs):
    s0, s1 = b * bs, min(E, (b + 1) * bs); B = s1 - s0
    print('batch', b, 'B', B)
    nodes = np.concatenate([src[s0:s1], dst[s0:s1], neg[s0:s1]]).astype(np.int32)
    nd = torch.from_numpy(nodes).cuda(); ed = torch.from_numpy(eidx[s0:s1].astype(np.int64)).cuda()
    tok = f.plan_device(nd, ed, 3, -1)
    wo = np.zeros(3 * B, np.int32); pf = np.zeros_like(wo); hv = np.zeros_like(wo); own = np.zeros(B, np.int32)
    cn = np.zeros(16, np.int32); cl = np.zeros(16, np.int32); ce = np.zeros((16, 2048), np.int32); nc = np.zeros(1, np.int32)
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = 0 if DBG else hooks.zt_test_tppr_plan_dump(f._live.h, P(wo), P(pf), P(hv), P(own), P(cn), P(cl), P(ce), P(nc))
    print("rc", rc, "chains", nc[0], "nodes", cn[:nc[0]], "len", cl[:nc[0]])
    u, v, g = nodes[:B], nodes[B:2 * B], nodes[2 * B:]
    # numpy restatement
    cnt = {}
    for i in range(B):
        for x in {int(u[i]), int(v[i]), int(g[i])}:
            cnt[x] = cnt.get(x, 0) + 1
    wr = {}
    wo_ref = np.zeros(3 * B, np.int32)
    for i in range(B):
        for r, x in enumerate((int(u[i]), int(v[i]), int(g[i]))):
            shadow = (r >= 1 and x == int(u[i])) or (r == 2 and x == int(v[i]))
            wo_ref[r * B + i] = 0 if shadow else wr.get(x, 0)
        for x in {int(u[i]), int(v[i])}:
            wr[x] = wr.get(x, 0) + 1
    print("wo equal:", np.array_equal(wo, wo_ref))
    bad = 0
    for c in range(nc[0]):
        x = cn[c]
        edges = [i for i in range(B) if u[i] == x or v[i] == x]
        if len(edges) != cl[c] or list(ce[c, :cl[c]]) != edges:
            bad += 1; print("chain", c, "node", x, "len", cl[c], "expected", len(edges), list(ce[c, :min(cl[c], 10)]), edges[:10])
    print("bad chains", bad)
    chain_of = {int(cn[c]): c for c in range(nc[0])}
    hv_ref = np.full(3 * B, -1, np.int32)
    for i in range(B):
        for r, x in enumerate((int(u[i]), int(v[i]), int(g[i]))):
            shadow = (r >= 1 and x == int(u[i])) or (r == 2 and x == int(v[i]))
            if shadow or x not in chain_of: continue
            c = chain_of[x]
            if cl[c] > 0: hv_ref[r * B + i] = c
    print("hv equal:", np.array_equal(hv, hv_ref), "mismatches", int((hv != hv_ref).sum()))
    own_ref = np.full(B, -1, np.int32)
    for i in range(B):
        a, b = int(u[i]), int(v[i])
        ia, ib = a in chain_of, (b in chain_of and b != a)
        if ia and (not ib or cnt[a] >= cnt[b]): own_ref[i] = chain_of[a]
        elif ib: own_ref[i] = chain_of[b]
    print("owner equal:", np.array_equal(own, own_ref))
    nodes3 = torch.from_numpy(nodes).cuda()
    try:
        f.stream_device(nd, torch.from_numpy(ts[s0:s1].astype(np.float64)).cuda(), ed, 3, True, -1, plan_token=tok)
        print("launch ok")
    except Exception as ex:
        print("launch failed:", str(ex)[:600])
    
