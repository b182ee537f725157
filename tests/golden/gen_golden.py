#!/usr/bin/env python3
"""Generate the golden vectors in tests/golden/*.npz from the reference itself.

Runs ONLY in the build container (needs /root/reference).  It imports the
reference's Python source unmodified under oracle/numba_standin (numba cannot
be installed here) and patches numba's two arithmetic specialities
(np.argsort quicksort, pow(float,int)) into the reference module's namespace
from oracle/numba_semantics.py.  The reference never leaves the container:
only the resulting data (inputs + expected outputs) is committed.

    python tests/golden/gen_golden.py [--check]

--check regenerates in memory and compares with the committed files.
"""
import argparse
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("ZEBRA_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(ROOT, "oracle", "numba_standin"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, HERE)
sys.path.insert(0, REF)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import inputs as I  # noqa: E402
import numba_semantics as ns  # noqa: E402

import utils.util as U  # noqa: E402  (reference)
U.np = ns.NumpyWithNumbaArgsort(np)
U.pow = ns.numba_int_pow
from model.tgn_model import TGN  # noqa: E402  (reference)
from model.time_encoding import TimeEncode  # noqa: E402  (reference)

OUT = {}


def save(name, **arrays):
    OUT[name] = arrays


def ref_state(f, m):
    """Dict items of model m in iteration order -> dense arrays."""
    N, k = f.num_nodes, f.k
    ln = np.zeros(N, np.int32)
    e = np.zeros((N, k), np.int64)
    nd = np.zeros((N, k), np.int64)
    ts = np.zeros((N, k), np.float64)
    w = np.zeros((N, k), np.float64)
    for v in range(N):
        d = f.PPR_list[m][v]
        ln[v] = len(d)
        for j, (key, val) in enumerate(d.items()):
            e[v, j], nd[v, j], ts[v, j], w[v, j] = key[0], key[1], key[2], val
    return dict(len=ln, norm=np.asarray(f.norm_list[m], np.float64).copy(), eidx=e, node=nd, ts=ts, w=w)


# ---------------------------------------------------------------------------
# G1 streaming / G2 fill + aliasing
# ---------------------------------------------------------------------------
def gen_stream():
    for name, (kind, N, E, seed, bs, k, al, be, full) in I.STREAM_CASES.items():
        src, dst, neg, ts, eidx = I.make_stream(kind, N, E, seed)
        f = U.tppr_finder(N, k, len(al), list(al), list(be))
        digests = []
        keep = {}
        nb = (E + bs - 1) // bs
        full_idx = list(range(nb)) if full else sorted({0, nb - 1})
        for b in range(nb):
            s, e_ = b * bs, min(E, (b + 1) * bs)
            nodes = np.concatenate([src[s:e_], dst[s:e_], neg[s:e_]]).astype(np.int32)
            t3 = np.concatenate([ts[s:e_]] * 3)
            on, oe, od, ow = f.streaming_topk(nodes, t3, eidx[s:e_])
            arrs = []
            for m in range(len(al)):
                assert on[m].dtype == np.int32 and oe[m].dtype == np.int32
                assert od[m].dtype == np.float32 and ow[m].dtype == np.float32
                arrs += [on[m], oe[m], od[m], ow[m]]
            digests.append(I.digest(arrs))
            if b in full_idx:
                keep["b%d_nodes" % b] = np.stack(on)
                keep["b%d_eidx" % b] = np.stack(oe)
                keep["b%d_dt" % b] = np.stack(od)
                keep["b%d_w" % b] = np.stack(ow)
        st = {}
        for m in range(len(al)):
            for kk, v in ref_state(f, m).items():
                st["state%d_%s" % (m, kk)] = v
        save("g1_stream_" + name, digests=np.array(digests), full_idx=np.array(full_idx), **keep, **st)

    # variants (utils/util.py:581-782) on one case: no_fake and single
    kind, N, E, seed, bs, k, al, be, _ = I.STREAM_CASES["tiny_general"]
    src, dst, neg, ts, eidx = I.make_stream(kind, N, E, seed)
    f = U.tppr_finder(N, k, len(al), list(al), list(be))
    n2 = np.concatenate([src[:40], dst[:40]]).astype(np.int32)
    on, oe, od, ow = f.streaming_topk_no_fake(n2, np.concatenate([ts[:40]] * 2), eidx[:40])
    n3 = np.concatenate([src[40:80], dst[40:80], neg[40:80]]).astype(np.int32)
    sn, se, sd, sw = f.single_streaming_topk(n3, np.concatenate([ts[40:80]] * 3), eidx[40:80], 1)
    st = {}
    for m in range(len(al)):
        for kk, v in ref_state(f, m).items():
            st["state%d_%s" % (m, kk)] = v
    save("g1_variants", nf_nodes=np.stack(on), nf_eidx=np.stack(oe), nf_dt=np.stack(od), nf_w=np.stack(ow),
         single_nodes=sn, single_eidx=se, single_dt=sd, single_w=sw, **st)

    # G2: compute_val_tppr == update-only replay; backup/restore aliasing
    kind, N, E, seed, bs, k, al, be, _ = I.STREAM_CASES["bip_k20"]
    src, dst, neg, ts, eidx = I.make_stream(kind, N, E, seed)
    f = U.tppr_finder(N, k, len(al), list(al), list(be))
    f.compute_val_tppr(src[:1000].astype(np.int64), dst[:1000].astype(np.int64), ts[:1000], eidx[:1000])
    st = {}
    for m in range(len(al)):
        for kk, v in ref_state(f, m).items():
            st["state%d_%s" % (m, kk)] = v
    backup = f.backup_tppr()
    alias = backup[1][0] is f.PPR_list[0]           # shallow copy: inner list shared
    nodes = np.concatenate([src[1000:1200], dst[1000:1200], neg[1000:1200]]).astype(np.int32)
    f.streaming_topk(nodes, np.concatenate([ts[1000:1200]] * 3), eidx[1000:1200])
    after = ref_state(f, 0)
    f.restore_tppr(backup)
    restored = ref_state(f, 0)
    restore_is_noop = all(np.array_equal(after[kk], restored[kk]) for kk in after)
    save("g2_fill", alias=np.array(alias), restore_is_noop=np.array(restore_is_noop), **st)


# ---------------------------------------------------------------------------
# G3 pruning
# ---------------------------------------------------------------------------
def gen_prune():
    for name, (kind, N, E, seed, nq, width, depth, k, alpha, beta) in I.PRUNE_CASES.items():
        src, dst, neg, ts, eidx = I.make_stream(kind, N, E, seed)
        data = types.SimpleNamespace(sources=src, destinations=dst, edge_idxs=eidx, timestamps=ts)
        nf = U.get_neighbor_finder(data)
        qn, qt = I.prune_queries(src, dst, ts, N, nq, seed)
        # ids beyond the adjacency length are an IndexError in the reference
        qn = np.minimum(qn, len(nf.node_to_neighbors) - 1).astype(np.int32)
        on = np.zeros((nq, k), np.int32)
        oe = np.zeros((nq, k), np.int32)
        od = np.zeros((nq, k), np.float32)
        ow = np.zeros((nq, k), np.float32)
        nf.get_pruned_topk(qn, qt, width, depth, alpha, beta, k, on, oe, od, ow)
        # adjacency of a few nodes, to pin the CSR builder's tie order
        probe = np.unique(qn)[:8]
        adj = {}
        for v in probe:
            adj["adj%d_nbr" % v] = nf.node_to_neighbors[v]
            adj["adj%d_eid" % v] = nf.node_to_edge_idxs[v]
            adj["adj%d_ts" % v] = nf.node_to_edge_timestamps[v]
        save("g3_prune_" + name, q_nodes=qn, q_ts=qt, nodes=on, eidx=oe, dt=od, w=ow, probe=probe, **adj)


# ---------------------------------------------------------------------------
# G4 embedding / G5 memory protocol / G6 TimeEncode
# ---------------------------------------------------------------------------
def build_tgn(N, E1, D, F, T, k, al, be, w, efeat, strategy="streaming", nf=None, width=10, depth=2):
    args = types.SimpleNamespace(alpha_list=list(al), beta_list=list(be), topk=k, tppr_strategy=strategy,
                                 n_degree=width, n_layer=depth, n_nodes=N, n_edges=E1)
    tgn = TGN(neighbor_finder=nf, node_features=None, edge_features=efeat.astype(np.float64), device="cpu",
              n_layers=depth, n_heads=2, dropout=0.0, use_memory=True, node_dimension=D, time_dimension=T,
              memory_dimension=D, embedding_module_type="diffusion", message_function="identity",
              aggregator_type="last", memory_updater_type="gru", n_neighbors=width, args=args)
    em = tgn.embedding_module
    with torch.no_grad():
        for mod, pre in ((em.fc1, "fc1"), (em.fc2, "fc2"), (em.fc1_source, "fc1s"), (em.fc2_source, "fc2s")):
            mod.weight.copy_(torch.from_numpy(w[pre + "_w"]))
            mod.bias.copy_(torch.from_numpy(w[pre + "_b"]))
        g = tgn.memory_updater.memory_updater
        g.weight_ih.copy_(torch.from_numpy(w["w_ih"]))
        g.weight_hh.copy_(torch.from_numpy(w["w_hh"]))
        g.bias_ih.copy_(torch.from_numpy(w["b_ih"]))
        g.bias_hh.copy_(torch.from_numpy(w["b_hh"]))
        tgn.affinity_score.fc1.weight.copy_(torch.from_numpy(w["aff1_w"]))
        tgn.affinity_score.fc1.bias.copy_(torch.from_numpy(w["aff1_b"]))
        tgn.affinity_score.fc2.weight.copy_(torch.from_numpy(w["aff2_w"]))
        tgn.affinity_score.fc2.bias.copy_(torch.from_numpy(w["aff2_b"]))
    em.drop.p = 0.0
    tgn.reset_timer()
    return tgn


def mem_state(tgn, pre):
    m = tgn.memory
    return {pre + "memory": m.memory.detach().numpy().copy(), pre + "last_update": m.last_update.numpy().copy(),
            pre + "messages": m.messages.detach().numpy().copy(), pre + "timestamps": m.timestamps.numpy().copy(),
            pre + "flags": m.nodes.astype(np.uint8).copy()}


def gen_embed():
    torch.set_num_threads(1)
    for name, (N, E, D, F, T, k, al, be, seed, bs, nb) in I.EMBED_CASES.items():
        src, dst, neg, ts, eidx = I.make_stream("general", N, E, seed)
        w = I.model_weights(D, F, T, len(al), seed)
        mem0, efeat = I.random_tables(N, E + 1, D, F, seed)
        tw = I.time_encode_weights(T)

        # --- G4: embedding alone, random memory table, eval mode -------------
        tgn = build_tgn(N, E + 1, D, F, T, k, al, be, w, efeat).eval()
        assert np.array_equal(tgn.time_encoder.w.weight.detach().numpy().ravel(), tw)
        # warm the T-PPR state on the first half, then embed one batch
        half = E // 2
        tgn.embedding_module.tppr_finder.compute_val_tppr(src[:half].astype(np.int64), dst[:half].astype(np.int64),
                                                          ts[:half], eidx[:half])
        tgn.embedding_module.tppr_finder.restore_val_tppr()
        s, e_ = half, half + bs
        nodes = np.concatenate([src[s:e_], dst[s:e_], neg[s:e_]])
        t3 = np.concatenate([ts[s:e_]] * 3)
        with torch.no_grad():
            emb = tgn.embedding_module.compute_embedding_tppr_ensemble(
                memory=torch.from_numpy(mem0), source_nodes=nodes, timestamps=t3, edge_idxs=eidx[s:e_],
                memory_updater=tgn.memory_updater, train=False)
        g4 = dict(emb=emb.numpy().copy(), average_topk=np.array(tgn.embedding_module.average_topk))

        # --- G5: 5 consecutive batches of the full protocol -------------------
        out = {}
        for mode, train in (("eval", False), ("train", True)):
            tgn = build_tgn(N, E + 1, D, F, T, k, al, be, w, efeat)
            tgn.train(train)
            for b in range(nb):
                s, e_ = b * bs, (b + 1) * bs
                ctx = torch.enable_grad() if train else torch.no_grad()
                with ctx:
                    se, de, ne = tgn.compute_temporal_embeddings(src[s:e_], dst[s:e_], neg[s:e_], ts[s:e_],
                                                                 eidx[s:e_], 10, train)
                    score = tgn.affinity_score(torch.cat([se, se], dim=0), torch.cat([de, ne])).squeeze(dim=0)
                    prob = score.sigmoid()
                out["%s_b%d_emb" % (mode, b)] = torch.cat([se, de, ne]).detach().numpy().copy()
                out["%s_b%d_prob" % (mode, b)] = prob.detach().numpy().copy().ravel()
                if train:
                    tgn.memory.detach_memory()
                if b == nb - 1:
                    out.update(mem_state(tgn, "%s_b%d_" % (mode, b)))
        save("g45_embed_" + name, **g4, **out)

    # --- G6 TimeEncode -------------------------------------------------------
    te = TimeEncode(100)
    dts = np.array([0.0, 1.0, 29.5, 86400.0, 1.2345e6, 2.4e8, 3.0e8, 16777216.0, 16777217.0, 0.001,
                    1e-9, 7.77e7, 123456.789, 5e5, 999999.0, 31.0], np.float32).reshape(4, 4)
    with torch.no_grad():
        enc = te(torch.from_numpy(dts)).numpy()
    save("g6_timeencode", time_w=te.w.weight.detach().numpy().ravel().copy(), dts=dts, enc=enc)


# ---------------------------------------------------------------------------
# G8 training step: loss and parameter gradients of train.py:205-215 (no optimizer step)
# ---------------------------------------------------------------------------
def gen_train_grads():
    torch.set_num_threads(1)
    name = "d20_f7"
    N, E, D, F, T, k, al, be, seed, bs, nb = I.EMBED_CASES[name]
    src, dst, neg, ts, eidx = I.make_stream("general", N, E, seed)
    w = I.model_weights(D, F, T, len(al), seed)
    _, efeat = I.random_tables(N, E + 1, D, F, seed)
    tgn = build_tgn(N, E + 1, D, F, T, k, al, be, w, efeat)
    tgn.train(True)
    crit = torch.nn.BCELoss()
    out = {}
    for b in range(nb):
        s, e_ = b * bs, (b + 1) * bs
        tgn.zero_grad()
        pos, negp = tgn.compute_edge_probabilities(src[s:e_], dst[s:e_], neg[s:e_], ts[s:e_], eidx[s:e_], 10, True)
        loss = crit(pos.squeeze(), torch.ones(bs)) + crit(negp.squeeze(), torch.zeros(bs))
        loss.backward()
        out["b%d_loss" % b] = np.float64(loss.item())
        for pn, p in tgn.named_parameters():
            if p.requires_grad and p.grad is not None:
                out["b%d_grad_%s" % (b, pn)] = p.grad.detach().numpy().copy()
        tgn.memory.detach_memory()
    save("g8_train_grads", **out)


# ---------------------------------------------------------------------------
# G7 ingest: the reference's get_data / compute_time_statistics on a synthetic ml_*.csv
# ---------------------------------------------------------------------------
def gen_ingest():
    import tempfile
    import utils.data_processing as DP  # noqa: E402  (reference)
    u, i, ts, label, idx = I.make_ml_table()
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, "data", "synth"))
        os.makedirs(os.path.join(tmp, "run"))
        I.write_ml_csv(os.path.join(tmp, "data", "synth", "ml_synth.csv"), u, i, ts, label, idx)
        cwd = os.getcwd()
        os.chdir(os.path.join(tmp, "run"))          # the reference reads '../data/{name}/ml_{name}.csv'
        try:
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                full, train, val, test, nn_val, nn_test, n_nodes, n_edges = DP.get_data("synth")
        finally:
            os.chdir(cwd)
    out = dict(u=u, i=i, ts=ts, label=label, idx=idx, n_nodes=np.int64(n_nodes), n_edges=np.int64(n_edges))
    for nm, d in (("full", full), ("train", train), ("val", val), ("test", test), ("nn_val", nn_val), ("nn_test", nn_test)):
        out[nm + "_idx"] = np.asarray(d.edge_idxs)
        out[nm + "_n_unique"] = np.int64(d.n_unique_nodes)
    out["time_stats"] = np.asarray(DP.compute_time_statistics(full.sources, full.destinations, full.timestamps), np.float64)
    save("g7_ingest", **out)


# ---------------------------------------------------------------------------
# G9 attention: the reference's own TemporalAttentionLayer (model/temporal_attention.py:7-68)
# ---------------------------------------------------------------------------
def gen_attention():
    from model.temporal_attention import TemporalAttentionLayer  # noqa: E402  (reference)
    torch.set_num_threads(1)
    for name, (D, F, T, k, N, heads, seed) in I.ATTENTION_CASES.items():
        w = I.attention_weights(D, F, T, seed)
        layer = TemporalAttentionLayer(D, D, F, T, output_dimension=D, n_head=heads, dropout=0.1).eval()
        mha = layer.multi_head_target
        with torch.no_grad():
            mha.q_proj_weight.copy_(torch.from_numpy(w["q_w"]))
            mha.k_proj_weight.copy_(torch.from_numpy(w["k_w"]))
            mha.v_proj_weight.copy_(torch.from_numpy(w["v_w"]))
            mha.in_proj_bias.copy_(torch.from_numpy(w["in_b"]))
            mha.out_proj.weight.copy_(torch.from_numpy(w["out_w"]))
            mha.out_proj.bias.copy_(torch.from_numpy(w["out_b"]))
            layer.merger.fc1.weight.copy_(torch.from_numpy(w["m1_w"]))
            layer.merger.fc1.bias.copy_(torch.from_numpy(w["m1_b"]))
            layer.merger.fc2.weight.copy_(torch.from_numpy(w["m2_w"]))
            layer.merger.fc2.bias.copy_(torch.from_numpy(w["m2_b"]))
        src, src_t, nbr, nbr_t, edge, mask = I.attention_inputs(D, F, T, k, N, seed)
        with torch.no_grad():
            out, aw = layer(torch.from_numpy(src), torch.from_numpy(src_t), torch.from_numpy(nbr),
                            torch.from_numpy(nbr_t), torch.from_numpy(edge), torch.from_numpy(mask.copy()))
        save("g9_attention_" + name, out=out.numpy().copy(), attn_w=aw.numpy().copy())


# ---------------------------------------------------------------------------
# G4/G5 with tppr_strategy='pruning' (modules/embedding_module.py:221-224,280-297; config C4's shape)
# ---------------------------------------------------------------------------
def gen_prune_embed():
    torch.set_num_threads(1)
    for name, (kind, N, E, D, F, T, k, al, be, width, depth, seed, bs, nb, first, n_train) in I.PRUNE_EMBED_CASES.items():
        src, dst, neg, ts, eidx = I.make_stream(kind, N, E, seed)
        w = I.model_weights(D, F, T, len(al), seed)
        mem0, efeat = I.random_tables(N, E + 1, D, F, seed)
        full = types.SimpleNamespace(sources=src, destinations=dst, edge_idxs=eidx, timestamps=ts)
        part = types.SimpleNamespace(sources=src[:n_train], destinations=dst[:n_train], edge_idxs=eidx[:n_train],
                                     timestamps=ts[:n_train])
        out = {}
        for mode, train in (("eval", False), ("train", True)):
            nf_full, nf_part = U.get_neighbor_finder(full), U.get_neighbor_finder(part)
            tgn = build_tgn(N, E + 1, D, F, T, k, al, be, w, efeat, strategy="pruning", nf=nf_part, width=width,
                            depth=depth)
            tgn.train(train)
            with torch.no_grad():
                tgn.memory.memory.copy_(torch.from_numpy(mem0))
            for b in range(nb):
                if b == nb // 2:
                    tgn.set_neighbor_finder(nf_full)            # the finder swap of train.py:191,245
                s, e_ = first + b * bs, first + (b + 1) * bs
                ctx = torch.enable_grad() if train else torch.no_grad()
                with ctx:
                    se, de, ne = tgn.compute_temporal_embeddings(src[s:e_], dst[s:e_], neg[s:e_], ts[s:e_],
                                                                 eidx[s:e_], width, train)
                    score = tgn.affinity_score(torch.cat([se, se], dim=0), torch.cat([de, ne])).squeeze(dim=0)
                    prob = score.sigmoid()
                out["%s_b%d_emb" % (mode, b)] = torch.cat([se, de, ne]).detach().numpy().copy()
                out["%s_b%d_prob" % (mode, b)] = prob.detach().numpy().copy().ravel()
                if train:
                    tgn.memory.detach_memory()
                if b == nb - 1:
                    out.update(mem_state(tgn, "%s_b%d_" % (mode, b)))
            out["%s_average_topk" % mode] = np.array(tgn.embedding_module.average_topk)
        save("g45_prune_" + name, **out)


# ---------------------------------------------------------------------------
# G10 epoch protocol (train.py:188-191,241-269,296-306), streaming strategy
# ---------------------------------------------------------------------------
def tppr_state(tgn, pre):
    f = tgn.embedding_module.tppr_finder
    st = {}
    for m in range(f.n_tppr):
        for kk, v in ref_state(f, m).items():
            st["%sm%d_%s" % (pre, m, kk)] = v
    return st


def gen_epoch():
    torch.set_num_threads(1)
    for name, (N, E, D, F, T, k, al, be, seed, bs, n_train, n_val, n_nn) in I.EPOCH_CASES.items():
        src, dst, neg, ts, eidx = I.make_stream("general", N, E, seed)
        w = I.model_weights(D, F, T, len(al), seed)
        _, efeat = I.random_tables(N, E + 1, D, F, seed)
        tgn = build_tgn(N, E + 1, D, F, T, k, al, be, w, efeat)
        em = tgn.embedding_module
        out = {}
        tr = slice(0, n_train)
        va = np.arange(n_train, n_train + n_val)
        nn_va = va[1::2][:n_nn]                      # a time-ordered subset, like new_node_val_data
        te = np.arange(n_train + n_val, E)

        def run(tag, idx, train):
            probs = []
            for s in range(0, len(idx), bs):
                ii = idx[s:s + bs]
                tgn.train(train)
                ctx = torch.enable_grad() if train else torch.no_grad()
                with ctx:
                    pos, negp = tgn.compute_edge_probabilities(src[ii], dst[ii], neg[ii], ts[ii], eidx[ii], 10, train)
                probs.append(torch.cat([pos, negp]).detach().numpy().ravel().copy())
            out[tag + "_prob"] = np.concatenate(probs)

        tppr_filled = False
        for epoch in range(2):
            ep = "e%d_" % epoch
            tgn.memory.__init_memory__()
            em.reset_tppr()
            tgn.set_neighbor_finder(None)
            run(ep + "train", np.arange(n_train), True)
            out.update(mem_state(tgn, ep + "train_end_"))
            em.reset_tppr()
            em.fill_tppr(src[tr].astype(np.int64), dst[tr].astype(np.int64), ts[tr], eidx[tr], tppr_filled)
            tppr_filled = True
            out.update(tppr_state(tgn, ep + "filled_"))
            train_memory_backup = tgn.memory.backup_memory()
            train_tppr_backup = em.backup_tppr()
            run(ep + "val", va, False)
            val_memory_backup = tgn.memory.backup_memory()
            val_tppr_backup = em.backup_tppr()
            tgn.memory.restore_memory(train_memory_backup)
            em.restore_tppr(train_tppr_backup)
            out[ep + "flags_after_restore_train"] = tgn.memory.nodes.astype(np.uint8).copy()
            out.update(tppr_state(tgn, ep + "after_restore_train_"))
            run(ep + "nn_val", nn_va, False)
            tgn.memory.restore_memory(val_memory_backup)
            em.restore_tppr(val_tppr_backup)
            out.update(mem_state(tgn, ep + "end_"))
            out.update(tppr_state(tgn, ep + "end_"))
        # test pass (train.py:296-306)
        val_memory_backup = tgn.memory.backup_memory()
        val_tppr_backup = em.backup_tppr()
        run("test", te, False)
        tgn.memory.restore_memory(val_memory_backup)
        em.restore_tppr(val_tppr_backup)
        out.update(mem_state(tgn, "final_"))
        out.update(tppr_state(tgn, "final_"))
        save("g10_epoch_" + name, **out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    a = ap.parse_args()
    gen_stream()
    gen_prune()
    gen_embed()
    gen_ingest()
    gen_train_grads()
    gen_attention()
    gen_prune_embed()
    gen_epoch()
    bad = 0
    if a.check:
        # the tie order of every fixture comes from numba_semantics.numba_argsort: pin it (and the C oracle's
        # zo_numba_argsort) to numba's own quicksort.py, loaded standalone (oracle/numba_pin.py)
        import numba_pin
        import pyoracle
        n_arr, n_bad = numba_pin.compare(extra=[pyoracle.numba_argsort])
        print("numba pin: %d tie-heavy arrays against %s, %d mismatches" % (n_arr, numba_pin.quicksort_path(), n_bad))
        bad += n_bad
    for name, arrays in OUT.items():
        path = os.path.join(HERE, name + ".npz")
        if a.check:
            old = np.load(path)
            for kk, v in arrays.items():
                if not np.array_equal(old[kk], np.asarray(v)):
                    print("MISMATCH", name, kk)
                    bad += 1
        else:
            np.savez_compressed(path, **arrays)
            print("%-32s %7.1f KB" % (name, os.path.getsize(path) / 1024))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
