"""Randomised parity soak of the streaming T-PPR kernel against the CPU oracle (test infrastructure, not collected by
pytest: run on a GPU box with   python tests/soak_tppr.py [seconds] [first seed]).
Every configuration draws its own graph shape (hub-dominated, dense, bipartite-like, power law), k, batch size (up to
launches that exercise several chains per model), alpha / beta (0.5 and 0.25 scale exactly: ties), self-loops, negatives
equal to endpoints or hubs, repeated timestamps; every batch's four output arrays and the final state must be bit-identical."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def one(seed, tppr, pyoracle):
    rng = np.random.RandomState(seed)
    N = int(rng.choice([12, 41, 300, 5000, 100000]))
    k = int(rng.choice([3, 5, 10, 20, 20, 20, 31]))
    bs = int(rng.choice([64, 600, 2048, 4096, 4096, 8192, 16384]))
    nb = int(rng.choice([2, 3, 5, 8]))
    M = int(rng.choice([1, 2, 2]))
    al = [float(rng.choice([0.0, 0.1, 0.2])) for _ in range(M)]
    be = [float(rng.choice([0.5, 0.5, 0.25, 0.8, 0.95])) for _ in range(M)]
    expo = float(rng.choice([0.0, 0.7, 1.0, 1.3]))
    E = bs * nb
    p = 1.0 / np.arange(1, N) ** expo
    p /= p.sum()
    perm = rng.permutation(N - 1) + 1
    src = perm[rng.choice(N - 1, E, p=p)].astype(np.int32)
    dst = perm[rng.choice(N - 1, E, p=p)].astype(np.int32)
    if rng.random_sample() < 0.5:                       # a star: one endpoint of many edges is THE hub
        hub = int(perm[0])
        m = rng.random_sample(E) < rng.choice([0.05, 0.2, 0.5])
        side = rng.random_sample(E) < 0.5
        src[m & side] = hub
        dst[m & ~side] = hub
    loops = rng.random_sample(E) < rng.choice([0.0, 0.01, 0.05])
    dst[loops] = src[loops]
    neg = perm[rng.choice(N - 1, E, p=p)].astype(np.int32)
    same = rng.random_sample(E) < rng.choice([0.0, 0.05])
    neg[same] = src[same]
    ts = np.cumsum(rng.randint(0, int(rng.choice([2, 3, 50])), E)).astype(np.float64)
    eidx = np.arange(1, E + 1, dtype=np.int64)
    if os.environ.get("ZT_SOAK_K"):
        k = int(os.environ["ZT_SOAK_K"])
    if os.environ.get("ZT_SOAK_MAX_N"):                  # (round 6, with a wide ZT_SOAK_K: dense rows, ties over 2k + 1 candidates,
        cap = int(os.environ["ZT_SOAK_MAX_N"])           #  in a few hundred edges -- the wide path is ~100 us per edge)
        if N > cap or bs * nb > 40 * cap:
            N = min(N, cap)
            bs = min(bs, 10 * cap)
            nb = min(nb, 3)
            E = bs * nb
            p = 1.0 / np.arange(1, N) ** expo
            p /= p.sum()
            perm = rng.permutation(N - 1) + 1
            src = perm[rng.choice(N - 1, E, p=p)].astype(np.int32)
            dst = perm[rng.choice(N - 1, E, p=p)].astype(np.int32)
            neg = perm[rng.choice(N - 1, E, p=p)].astype(np.int32)
            ts = np.cumsum(rng.randint(0, 3, E)).astype(np.float64)
            eidx = np.arange(1, E + 1, dtype=np.int64)
    if os.environ.get("ZT_SOAK_VERBOSE"):
        print("seed %d: N=%d k=%d bs=%d nb=%d M=%d alpha=%s beta=%s expo=%.1f" % (seed, N, k, bs, nb, M, al, be, expo), flush=True)
    f = tppr.tppr_finder(N, k, M, al, be)
    o = pyoracle.TpprOracle(N, k, M, al, be)
    variants = rng.random_sample() < 0.4                 # the other entry points of the finder, batch by batch
    reload_at = int(rng.randint(1, nb)) * bs if rng.random_sample() < 0.3 else -1
    backup = None
    for s in range(0, E, bs):
        e = s + bs
        if s == reload_at:
            # checkpoint round trip in mid-stream: the rows of the touched nodes into a NEW finder (state_dict /
            # load_state_dict), which carries on; and a snapshot (backup_tppr) that the old finder must get back later
            touched = np.unique(np.concatenate([src[:s], dst[:s]])).astype(np.int64)
            sd = f.state_dict(touched)
            backup = (f, f.backup_tppr(), [f.export_state(m) for m in range(M)])
            f = tppr.tppr_finder(N, k, M, al, be)
            f.load_state_dict(sd)
        nodes = np.concatenate([src[s:e], dst[s:e], neg[s:e]])
        v = rng.randint(4) if variants else 0
        if v == 1:                                       # utils/util.py:682-782
            n2 = np.concatenate([src[s:e], dst[s:e]])
            a, b = f.streaming_topk_no_fake(n2, ts[s:e], eidx[s:e]), o.streaming_topk_no_fake(n2, ts[s:e], eidx[s:e])
        elif v == 2:                                     # utils/util.py:581-679, model after model
            a, b = [[] for _ in range(4)], [[] for _ in range(4)]
            for m in range(M):
                ga = f.single_streaming_topk(nodes, ts[s:e], eidx[s:e], m)
                gb = o.single_streaming_topk(nodes, ts[s:e], eidx[s:e], m)
                for q in range(4):
                    a[q].append(ga[q]); b[q].append(gb[q])
        elif v == 3:                                     # utils/util.py:787-873: update only, nothing emitted
            f.compute_val_tppr(src[s:e], dst[s:e], ts[s:e], eidx[s:e])
            o.update_only(src[s:e], dst[s:e], ts[s:e], eidx[s:e])
            continue
        else:
            a, b = f.streaming_topk(nodes, ts[s:e], eidx[s:e]), o.streaming_topk(nodes, ts[s:e], eidx[s:e])
        for x, y, nm in zip(a, b, ("nodes", "eidx", "dt", "w")):
            if not np.array_equal(np.stack(x), np.stack(y)):
                return "seed %d: %s differs in the batch at %d, entry point %d (N=%d k=%d bs=%d M=%d alpha=%s beta=%s expo=%.1f)" % (
                    seed, nm, s, v, N, k, bs, M, al, be, expo)
    for m in range(M):
        ga, wa = f.export_state(m), o.export(m)
        for kk in ("len", "norm", "eidx", "node", "ts", "w"):
            if not np.array_equal(ga[kk], wa[kk]):
                return "seed %d: final state %s of model %d differs (N=%d k=%d bs=%d)" % (seed, kk, m, N, k, bs)
    if backup is not None:                               # the old finder went on existing: restore its snapshot
        f0, snap, want0 = backup
        nodes = np.concatenate([src[:bs], dst[:bs], neg[:bs]])
        f0.streaming_topk(nodes, ts[E - bs:E] + 1.0, eidx[:bs] + E)        # (disturb it first)
        f0.restore_tppr(snap)
        for m in range(M):
            g0 = f0.export_state(m)
            for kk in ("len", "norm", "eidx", "node", "ts", "w"):
                if not np.array_equal(g0[kk], want0[m][kk]):
                    return "seed %d: state %s of model %d differs after backup_tppr / restore_tppr" % (seed, kk, m)
    return None


def main():
    import torch
    assert torch.cuda.is_available()
    if os.environ.get("ZT_SOAK_LIB"):                   # an experimental build of the library
        from zebra_amd import _capi
        _capi.LIB_PATH = os.environ["ZT_SOAK_LIB"]
    from zebra_amd import tppr
    import pyoracle
    pyoracle.lib()
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    t0, n = time.time(), 0
    while time.time() - t0 < budget:
        err = one(seed, tppr, pyoracle)
        if err:
            print("FAIL", err)
            sys.exit(1)
        n += 1
        seed += 1
        if n % 10 == 0:
            print("%d configurations bit-identical (%.0f s)" % (n, time.time() - t0), flush=True)
    print("soak ok: %d configurations, seeds up to %d" % (n, seed - 1))


if __name__ == "__main__":
    main()
