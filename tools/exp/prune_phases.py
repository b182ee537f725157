"""Times k_pruned_topk on a C4-shaped query batch with the kernel cut short after the walk / after the merge
(ZT_PRUNE_STOP, diagnostic): where does a query's time go?   python tools/exp/prune_phases.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT)
    import types, numpy as np, torch
    from zebra_amd import synth
    from zebra_amd.tppr import get_neighbor_finder
    wl = synth.WORKLOADS["c4"]
    E = 400000
    src, dst, ts, eidx = synth.power_law_stream(wl["n_nodes"], E, seed=5, perm_seed=7)
    nf = get_neighbor_finder(types.SimpleNamespace(sources=src, destinations=dst, edge_idxs=eidx, timestamps=ts))
    dev = torch.device("cuda")
    bs = 1000
    s0 = E - bs
    neg = synth.negatives(dst, E, seed=6)
    q = torch.from_numpy(np.concatenate([src[s0:], dst[s0:], neg[s0:]])).to(dev)
    t = torch.from_numpy(np.concatenate([ts[s0:]] * 3)).to(dev)
    k = 40
    outs = [torch.zeros((2, 3 * bs, k), dtype=dt, device=dev) for dt in (torch.int32, torch.int32, torch.float32, torch.float32)]
    for betas in ([0.5, 0.95], [0.5, 0.5], [0.95, 0.95]):
        al = [0.1] * len(betas)
        for _ in range(5):
            nf.pruned_topk_multi_device(q, t, 10, 2, al, betas, k, *outs, check_status=False)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50):
            nf.pruned_topk_multi_device(q, t, 10, 2, al, betas, k, *outs, check_status=False)
        b.record()
        torch.cuda.synchronize()
        print("ZT_PRUNE_STOP=%s betas %s: %.1f us per launch" % (os.environ.get("ZT_PRUNE_STOP", "0"), betas, 1e3 * a.elapsed_time(b) / 50))
    if os.environ.get("ZT_PRUNE_STOP") == "2":
        nd = outs[0][0, :, 0].cpu().numpy()
        print("candidates after the merge: mean %.1f, <=40: %.2f, <=64: %.2f, max %d" % (nd.mean(), (nd <= 40).mean(), (nd <= 64).mean(), nd.max()))
else:
    for stop in ("2", "4", "3", "0"):
        subprocess.run([sys.executable, __file__, "run"], env=dict(os.environ, ZT_PRUNE_STOP=stop), check=True)
