"""Randomised parity soak of the whole eval step (TGN.step_device through the native pipeline: grouped T-PPR launches,
CU-masked streams, the message build beside the aggregation, the projected table, the GRU) against the CPU oracle's
protocol -- test infrastructure, run on a GPU box:   python tests/soak_pipeline.py [seconds] [first seed]
Random node counts, batch sizes (a ragged last batch now and then), k, widths (the reference's 100/100 and others: every
aggregation kernel), feature widths, one or two models, launch groups 1..4, CU masks, views ahead of different lengths,
shuffled node ids.  Per batch: embeddings <= 1e-4; at the end: T-PPR state bit-exact, memory / messages <= 1e-4,
last_update / flags exact."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)
TOL = 1e-4


def one(seed, torch, pyoracle):
    import inputs as I
    from helpers import build_tgn
    from zebra_amd import synth
    rng = np.random.RandomState(seed)
    N = int(rng.choice([60, 2000, 100000]))
    bs = int(rng.choice([20, 200, 600, 1000, 4096]))
    nb = int(rng.randint(4, 11))
    k = int(rng.choice([5, 10, 20, 20, 40]))
    D, T = [(100, 100), (100, 100), (100, 100), (32, 16), (64, 100)][rng.randint(5)]
    F = int(rng.choice([1, 1, 4, 16, 172]))
    M = int(rng.choice([1, 2, 2]))
    al = [float(rng.choice([0.1, 0.2])) for _ in range(M)]
    be = [float(rng.choice([0.5, 0.8, 0.95])) for _ in range(M)]
    group = int(rng.choice([1, 2, 3, 4]))
    cus = int(rng.choice([0, 32, 64, 96]))
    look = int(rng.randint(0, 3 * group + 2))
    # (round 6; drawn AFTER the variables above so that the earlier rounds' seeds keep their configurations) how a launch group
    # reaches the aggregation: by member (the library's pick: the gate inside the aggregation kernel or as a one-wave kernel,
    # by batch size and CU masks) or by launch; now and then groups of 8
    rng2 = np.random.RandomState(seed + 777)
    by_launch = rng2.random_sample() < 0.25
    if rng2.random_sample() < 0.2:
        group, look = 8, int(rng2.randint(0, 27))
    ragged = rng.random_sample() < 0.3
    perm = None if rng.random_sample() < 0.5 else int(rng.randint(1, 100))
    strategy = "pruning" if rng.random_sample() < 0.3 else "streaming"
    width, depth = [(10, 2), (5, 2), (3, 3), (10, 1)][rng.randint(4)]
    E = bs * nb - (int(rng.randint(1, bs)) if ragged and bs > 1 else 0)
    tag = "seed %d: %s N=%d bs=%d nb=%d E=%d k=%d D=%d T=%d F=%d M=%d beta=%s group=%d cus=%d look=%d perm=%s width=%d depth=%d" % (
        seed, strategy, N, bs, nb, E, k, D, T, F, M, be, group, cus, look, perm, width, depth)
    tag += " release=%s" % ("launch" if by_launch else "member")
    if os.environ.get("ZT_SOAK_VERBOSE"):
        print(tag, flush=True)
    bip = (max(1, N // 3), N - max(1, N // 3)) if rng.randint(2) else None
    src, dst, ts, eidx = synth.power_law_stream(N, E, seed=seed, perm_seed=perm, bipartite=bip)
    neg = synth.negatives(dst, E, seed=seed + 1)
    w = I.model_weights(D, F, T, M, seed + 2)
    efeat = synth.edge_features(E + 1, F, seed=seed + 3)
    tw = I.time_encode_weights(T)
    dev = torch.device("cuda")
    t = [torch.from_numpy(x).to(dev) for x in (src, dst, neg, ts, eidx)]
    batches = [tuple(x[b * bs:min(E, (b + 1) * bs)] for x in t) for b in range(nb)]
    nf = ref_nf = None
    if strategy == "pruning":                       # the static graph of the whole stream (the reference's full_ngh_finder)
        import types
        from zebra_amd.tppr import get_neighbor_finder
        nf = get_neighbor_finder(types.SimpleNamespace(sources=src, destinations=dst, edge_idxs=eidx, timestamps=ts))
        ref_nf = pyoracle.CsrOracle(src, dst, eidx, ts, N + 1)
    tgn = build_tgn(N + 1, E + 1, D, F, T, k, al, be, w, efeat, strategy=strategy, nf=nf, width=width, depth=depth).eval()
    from zebra_amd import _capi
    _capi.set_kernel_choice(_capi.CHOICE_GROUP_RELEASE, _capi.RELEASE_LAUNCH if by_launch else 0)
    tgn.enable_pipeline(tppr_cus=cus, group=group, max_batch=max(bs, 64))
    p = pyoracle.ProtocolOracle(N + 1, D, F, T, k, al, be, w, efeat, tw, strategy, ref_nf, width, depth, n_threads=8)
    embs = []
    try:
        with torch.cuda.stream(tgn.main_stream):
            for b, cur in enumerate(batches):
                embs.append(tgn.step_device(*cur, ahead=batches[b + 1: b + 1 + look]).clone())
        torch.cuda.synchronize()
        if strategy == "streaming":
            tgn.embedding_module.tppr_finder.check_status()
        worst = 0.0
        for b in range(nb):
            s, e = b * bs, min(E, (b + 1) * bs)
            ref, _ = p.batch(src[s:e], dst[s:e], neg[s:e], ts[s:e], eidx[s:e], False)
            worst = max(worst, float(np.abs(embs[b].cpu().numpy() - ref).max()))
        if not worst <= TOL:
            return "%s: embeddings differ from the oracle by %g" % (tag, worst)
        f = tgn.embedding_module.tppr_finder if strategy == "streaming" else None
        for m in range(M if strategy == "streaming" else 0):
            a, bb = f.export_state(m), p.tppr.export(m)
            for kk in a:
                if not np.array_equal(a[kk], bb[kk]):
                    return "%s: T-PPR state %s of model %d differs" % (tag, kk, m)
        if np.abs(tgn.memory.memory.cpu().numpy() - p.mem.memory).max() > TOL:
            return "%s: memory differs" % tag
        if not np.array_equal(tgn.memory.last_update.cpu().numpy(), p.mem.last_update):
            return "%s: last_update differs" % tag
        if np.abs(tgn.memory.messages.cpu().numpy() - p.mem.messages).max() > TOL:
            return "%s: messages differ" % tag
        if not np.array_equal(tgn.memory.nodes.astype(np.uint8), p.mem.flags):
            return "%s: flags differ" % tag
    finally:
        torch.cuda.synchronize()
        tgn.enable_pipeline(False)
    return None


def main():
    import torch
    assert torch.cuda.is_available()
    import pyoracle
    pyoracle.lib()
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 9000
    t0, n = time.time(), 0
    while time.time() - t0 < budget:
        err = one(seed, torch, pyoracle)
        if err:
            print("FAIL", err)
            sys.exit(1)
        n += 1
        seed += 1
        if n % 5 == 0:
            print("%d configurations within tolerance (%.0f s)" % (n, time.time() - t0), flush=True)
    print("soak ok: %d configurations, seeds up to %d" % (n, seed - 1))


if __name__ == "__main__":
    main()
