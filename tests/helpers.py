"""Shared test helpers (own code)."""
import types

import numpy as np
import torch


def make_args(N, E1, k, al, be, strategy="streaming", width=10, depth=2, compat=False):
    return types.SimpleNamespace(alpha_list=list(al), beta_list=list(be), topk=k, tppr_strategy=strategy,
                                 n_degree=width, n_layer=depth, n_nodes=N, n_edges=E1,
                                 reference_compat_aliasing=compat)


def load_weights(tgn, w):
    em = tgn.embedding_module
    dev = tgn.device
    with torch.no_grad():
        for mod, pre in ((em.fc1, "fc1"), (em.fc2, "fc2"), (em.fc1_source, "fc1s"), (em.fc2_source, "fc2s")):
            mod.weight.copy_(torch.from_numpy(w[pre + "_w"]))
            mod.bias.copy_(torch.from_numpy(w[pre + "_b"]))
        g = tgn.memory_updater.memory_updater
        g.weight_ih.copy_(torch.from_numpy(w["w_ih"]))
        g.weight_hh.copy_(torch.from_numpy(w["w_hh"]))
        g.bias_ih.copy_(torch.from_numpy(w["b_ih"]))
        g.bias_hh.copy_(torch.from_numpy(w["b_hh"]))
        a = tgn.affinity_score
        a.fc1.weight.copy_(torch.from_numpy(w["aff1_w"]))
        a.fc1.bias.copy_(torch.from_numpy(w["aff1_b"]))
        a.fc2.weight.copy_(torch.from_numpy(w["aff2_w"]))
        a.fc2.bias.copy_(torch.from_numpy(w["aff2_b"]))
    em.drop.p = 0.0
    return tgn.to(dev)


def build_tgn(N, E1, D, F, T, k, al, be, w, efeat, strategy="streaming", nf=None, width=10, depth=2, compat=False):
    from zebra_amd.tgn import TGN
    args = make_args(N, E1, k, al, be, strategy, width, depth, compat)
    tgn = TGN(neighbor_finder=nf, node_features=None, edge_features=efeat, device="cuda", n_layers=depth,
              n_heads=2, dropout=0.0, use_memory=True, node_dimension=D, time_dimension=T, memory_dimension=D,
              embedding_module_type="diffusion", message_function="identity", aggregator_type="last",
              memory_updater_type="gru", n_neighbors=width, args=args)
    return load_weights(tgn.to("cuda"), w)


def epoch_protocol(p, g, I_case, streams, check, aliasing=True):
    """train.py:188-191,241-269,296-306 driven on any object with the drop-in's surface
    (``p``: adapter with init_memory/reset_tppr/fill_tppr/backup_*/restore_*/batch/state)."""
    N, E, D, F, T, k, al, be, seed, bs, n_train, n_val, n_nn = I_case
    src, dst, neg, ts, eidx = streams
    va = np.arange(n_train, n_train + n_val)
    nn_va = va[1::2][:n_nn]
    te = np.arange(n_train + n_val, E)

    def run(tag, idx, train):
        probs = [p.batch(src[idx[s:s + bs]], dst[idx[s:s + bs]], neg[idx[s:s + bs]], ts[idx[s:s + bs]],
                         eidx[idx[s:s + bs]], train) for s in range(0, len(idx), bs)]
        check(tag + "_prob", np.concatenate(probs))

    filled = False
    for epoch in range(2):
        ep = "e%d_" % epoch
        p.init_memory()
        p.reset_tppr()
        run(ep + "train", np.arange(n_train), True)
        check(ep + "train_end_", p.memory_state())
        p.reset_tppr()
        p.fill_tppr(src[:n_train], dst[:n_train], ts[:n_train], eidx[:n_train], filled)
        filled = True
        check(ep + "filled_", p.tppr_state())
        mb, tb = p.backup_memory(), p.backup_tppr()
        run(ep + "val", va, False)
        vmb, vtb = p.backup_memory(), p.backup_tppr()
        p.restore_memory(mb)
        p.restore_tppr(tb)
        if not aliasing:
            return                                  # deep snapshots part ways with the reference here
        check(ep + "flags_after_restore_train", p.memory_state()["flags"])
        check(ep + "after_restore_train_", p.tppr_state())
        run(ep + "nn_val", nn_va, False)
        p.restore_memory(vmb)
        p.restore_tppr(vtb)
        check(ep + "end_", p.memory_state())
        check(ep + "end_", p.tppr_state())
    vmb, vtb = p.backup_memory(), p.backup_tppr()
    run("test", te, False)
    p.restore_memory(vmb)
    p.restore_tppr(vtb)
    check("final_", p.memory_state())
    check("final_", p.tppr_state())


def make_checker(g, tol):
    def check(key, val):
        if isinstance(val, dict):
            for kk, v in val.items():
                check(key + kk, v)
            return
        want = g[key]
        if want.dtype.kind == "f" and want.dtype.itemsize == 4 and not key.endswith(("last_update", "timestamps")):
            assert np.abs(np.asarray(val, np.float32) - want).max() <= tol, key
        else:
            assert np.array_equal(np.asarray(val), want), key
    return check
