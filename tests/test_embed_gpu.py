"""GPU parity of the aggregation (P2) and memory-update (P3) kernels and of the
whole per-batch protocol, through the reference-shaped Python surface:
golden vectors (1e-4, BASELINE.json north_star tolerance) and the CPU oracle."""
import time

import numpy as np
import pytest
import torch

import inputs as I
from zebra_amd import _capi
from conftest import golden
from helpers import build_tgn

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.mark.parametrize("name", list(I.EMBED_CASES))
def test_embedding_golden(name):
    N, E, D, F, T, k, al, be, seed, bs, nb = I.EMBED_CASES[name]
    g = golden("g45_embed_" + name)
    src, dst, neg, ts, eidx = I.make_stream("general", N, E, seed)
    w = I.model_weights(D, F, T, len(al), seed)
    mem0, efeat = I.random_tables(N, E + 1, D, F, seed)
    tgn = build_tgn(N, E + 1, D, F, T, k, al, be, w, efeat).eval()
    assert np.array_equal(tgn.time_encoder.w.weight.detach().cpu().numpy().ravel(), I.time_encode_weights(T))
    em = tgn.embedding_module
    half = E // 2
    em.tppr_finder.compute_val_tppr(src[:half], dst[:half], ts[:half], eidx[:half])
    em.tppr_finder.restore_val_tppr()
    s, e = half, half + bs
    nodes = np.concatenate([src[s:e], dst[s:e], neg[s:e]])
    emb = em.compute_embedding_tppr_ensemble(memory=torch.from_numpy(mem0).cuda(), source_nodes=nodes,
                                             timestamps=np.concatenate([ts[s:e]] * 3), edge_idxs=eidx[s:e],
                                             memory_updater=tgn.memory_updater, train=False)
    assert emb.shape == g["emb"].shape
    assert np.abs(emb.cpu().numpy() - g["emb"]).max() <= TOL
    assert abs(em.average_topk - float(g["average_topk"])) < 1e-6


@pytest.mark.parametrize("mode", ["eval", "train"])
@pytest.mark.parametrize("name", list(I.EMBED_CASES))
def test_protocol_golden(name, mode):
    """Consecutive batches of TGN.compute_temporal_embeddings (tgn_model.py:124-174)."""
    N, E, D, F, T, k, al, be, seed, bs, nb = I.EMBED_CASES[name]
    g = golden("g45_embed_" + name)
    src, dst, neg, ts, eidx = I.make_stream("general", N, E, seed)
    w = I.model_weights(D, F, T, len(al), seed)
    _, efeat = I.random_tables(N, E + 1, D, F, seed)
    tgn = build_tgn(N, E + 1, D, F, T, k, al, be, w, efeat)
    train = mode == "train"
    tgn.train(train)
    for b in range(nb):
        s, e = b * bs, (b + 1) * bs
        ctx = torch.enable_grad() if train else torch.no_grad()
        with ctx:
            pos, negp = tgn.compute_edge_probabilities(src[s:e], dst[s:e], neg[s:e], ts[s:e], eidx[s:e], 10, train)
        prob = torch.cat([pos, negp]).detach().cpu().numpy().ravel()
        assert np.abs(prob - g["%s_b%d_prob" % (mode, b)]).max() <= TOL, "batch %d" % b
        if train:
            tgn.memory.detach_memory()
    m = tgn.memory
    pre = "%s_b%d_" % (mode, nb - 1)
    assert np.abs(m.memory.cpu().numpy() - g[pre + "memory"]).max() <= TOL
    assert np.array_equal(m.last_update.cpu().numpy(), g[pre + "last_update"])
    assert np.abs(m.messages.cpu().numpy() - g[pre + "messages"]).max() <= TOL
    assert np.array_equal(m.timestamps.cpu().numpy(), g[pre + "timestamps"])
    assert np.array_equal(m.nodes.astype(np.uint8), g[pre + "flags"])


def test_embeddings_golden_all_batches():
    """Embeddings of every eval batch (not only the probabilities)."""
    name = "d100_f172"
    N, E, D, F, T, k, al, be, seed, bs, nb = I.EMBED_CASES[name]
    g = golden("g45_embed_" + name)
    src, dst, neg, ts, eidx = I.make_stream("general", N, E, seed)
    w = I.model_weights(D, F, T, len(al), seed)
    _, efeat = I.random_tables(N, E + 1, D, F, seed)
    tgn = build_tgn(N, E + 1, D, F, T, k, al, be, w, efeat).eval()
    for b in range(nb):
        s, e = b * bs, (b + 1) * bs
        with torch.no_grad():
            se, de, ne = tgn.compute_temporal_embeddings(src[s:e], dst[s:e], neg[s:e], ts[s:e], eidx[s:e], 10, False)
        emb = torch.cat([se, de, ne]).cpu().numpy()
        assert np.abs(emb - g["eval_b%d_emb" % b]).max() <= TOL, "batch %d" % b


@pytest.mark.parametrize("cfg", [
    dict(N=3000, E=6000, F=172, bs=200, k=20, al=[0.1, 0.1], be=[0.5, 0.95], warm=4000, nb=6, seed=201),
    dict(N=5000, E=9000, F=1, bs=512, k=20, al=[0.1, 0.1], be=[0.5, 0.95], warm=5000, nb=5, seed=202),
    dict(N=800, E=4000, F=4, bs=100, k=40, al=[0.2], be=[0.8], warm=3000, nb=5, seed=203),
    # k beyond a wavefront (train.py:46 puts no bound on --topk): the wide streaming path (csrc/tppr_wide.hpp) feeding the
    # generic aggregation kernel, whose workgroup tile holds one query row of up to 80 neighbours
    dict(N=300, E=3000, F=1, bs=100, k=64, al=[0.1, 0.1], be=[0.5, 0.95], warm=2000, nb=5, seed=204),
    dict(N=200, E=2400, F=4, bs=80, k=80, al=[0.2], be=[0.5], warm=1600, nb=5, seed=205),
    # ... and beyond 80: the generic kernel's 16-tile instantiation over the projected table (one query row of up to 255)
    dict(N=150, E=3000, F=1, bs=100, k=128, al=[0.1, 0.1], be=[0.5, 0.95], warm=2400, nb=4, seed=206),
    dict(N=90, E=2700, F=4, bs=60, k=255, al=[0.2], be=[0.8], warm=2400, nb=4, seed=207),
])
def test_protocol_vs_oracle(oracle, cfg):
    """Larger seeded runs: T-PPR state bit-exact, embeddings / memory within 1e-4
    of the CPU oracle after several dependent batches."""
    D = T = 100
    N, E, F, bs, k, al, be = cfg["N"], cfg["E"], cfg["F"], cfg["bs"], cfg["k"], cfg["al"], cfg["be"]
    M = len(al)
    src, dst, neg, ts, eidx = I.make_stream("bipartite", N, E, cfg["seed"])
    w = I.model_weights(D, F, T, M, cfg["seed"])
    _, efeat = I.random_tables(N, E + 1, D, F, cfg["seed"])
    tw = I.time_encode_weights(T)
    tgn = build_tgn(N, E + 1, D, F, T, k, al, be, w, efeat).eval()
    f = oracle.TpprOracle(N, k, M, al, be)
    mem = oracle.MemoryOracle(N, D, 2 * D + F + T)
    gru = {kk: w[kk] for kk in ("w_ih", "w_hh", "b_ih", "b_hh")}
    s = 0
    while s < cfg["warm"] + cfg["nb"] * bs:
        e = s + bs
        nodes = np.concatenate([src[s:e], dst[s:e], neg[s:e]])
        with torch.no_grad():
            se, de, ne = tgn.compute_temporal_embeddings(src[s:e], dst[s:e], neg[s:e], ts[s:e], eidx[s:e], 10, False)
        on, oe, od, ow = f.streaming_topk(nodes, ts[s:e], eidx[s:e])
        emb = oracle.embed(mem.memory, efeat, tw, nodes, np.stack(on), np.stack(oe), np.stack(od), np.stack(ow), w,
                           n_threads=8)
        got = torch.cat([se, de, ne]).cpu().numpy()
        assert np.abs(got - emb).max() <= TOL, "embeddings differ at edge %d" % s
        mem.store_messages(efeat, tw, src[s:e], dst[s:e], ts[s:e], eidx[s:e])
        mem.gru_update(gru, np.unique(np.concatenate([src[s:e], dst[s:e]])), n_threads=8)
        s = e
    for m in range(M):
        a, b = tgn.embedding_module.tppr_finder.export_state(m), f.export(m)
        for kk in a:
            assert np.array_equal(a[kk], b[kk])
    assert np.abs(tgn.memory.memory.cpu().numpy() - mem.memory).max() <= TOL
    assert np.array_equal(tgn.memory.last_update.cpu().numpy(), mem.last_update)
    assert np.abs(tgn.memory.messages.cpu().numpy() - mem.messages).max() <= TOL
    assert np.array_equal(tgn.memory.nodes.astype(np.uint8), mem.flags)


@pytest.mark.parametrize("tppr_cus,strategy", [(0, "streaming"), (32, "streaming"), (0, "pruning")])
def test_pipelined_step_matches_sequential(tppr_cus, strategy):
    """The native step pipeline (T-PPR query of the next batch on a side stream, optionally CU-masked, the
    dependency prepass planned two batches ahead on a third stream, everything enqueued by one C call) must
    not change any result; nor must restricting the embedded rows to a shard."""
    import types
    from zebra_amd.tppr import get_neighbor_finder
    name = "d100_f1"
    N, E, D, F, T, k, al, be, seed, bs, nb = I.EMBED_CASES[name]
    src, dst, neg, ts, eidx = I.make_stream("general", N, E, seed)
    w = I.model_weights(D, F, T, len(al), seed)
    _, efeat = I.random_tables(N, E + 1, D, F, seed)
    dev = torch.device("cuda")
    t = [torch.from_numpy(x).to(dev) for x in (src, dst, neg, ts, eidx)]
    outs = {}
    for mode in ("seq", "pipe", "pipe_rows"):
        nf = None
        if strategy == "pruning":
            nf = get_neighbor_finder(types.SimpleNamespace(sources=src, destinations=dst, edge_idxs=eidx, timestamps=ts))
        tgn = build_tgn(N, E + 1, D, F, T, k, al, be, w, efeat, strategy=strategy, nf=nf).eval()
        if mode != "seq":
            tgn.enable_pipeline(tppr_cus=tppr_cus, max_batch=64)
        nbt = E // bs
        embs = []
        rows = (bs, 3 * bs - 7) if mode == "pipe_rows" else None
        main = getattr(tgn, "main_stream", None) or torch.cuda.current_stream()
        with torch.cuda.stream(main):
            for b in range(nbt):
                cur = tuple(x[b * bs:(b + 1) * bs] for x in t)
                nxt = tuple(x[(b + 1) * bs:(b + 2) * bs] for x in t) if (mode != "seq" and b + 1 < nbt) else None
                # every third batch is left unplanned: planned and inline prepasses interleave
                nx2 = tuple(x[(b + 2) * bs:(b + 3) * bs] for x in t) if (mode != "seq" and b + 2 < nbt and b % 3) else None
                embs.append(tgn.step_device(*cur, prefetch=nxt, plan=nx2, rows=rows).clone())
        torch.cuda.synchronize()
        state = {}
        if strategy == "streaming":
            tgn.embedding_module.tppr_finder.check_status()
            state = tgn.embedding_module.tppr_finder.export_state(0)
        outs[mode] = (torch.stack(embs).cpu().numpy(), tgn.memory.memory.cpu().numpy(), state)
        tgn.enable_pipeline(False)
    assert np.array_equal(outs["seq"][0], outs["pipe"][0])
    assert np.array_equal(outs["seq"][0][:, bs:3 * bs - 7], outs["pipe_rows"][0])
    for mode in ("pipe", "pipe_rows"):
        assert np.array_equal(outs["seq"][1], outs[mode][1])
        for kk in outs["seq"][2]:
            assert np.array_equal(outs["seq"][2][kk], outs[mode][2][kk])


@pytest.mark.parametrize("F,bs", [(1, 400), (172, 400), (1, 150)])
def test_fused_output_and_gru_launch_matches_separate_kernels(F, bs):
    """k_out_gru / k_out_gru2 (the output layers beside the GRU update in ONE launch, csrc/memory_update.hip): a pipelined
    step against the sequential path -- which launches k_embed_out / k_embed_out2 and k_gru / k_gru_split one after the
    other -- bit for bit: embeddings of every batch, the memory table, last_update.  bs = 400: 1 200 rows, the tiled output
    kernel + k_gru<1> (partial-sum groups with F = 172: k_out_gru<5>); bs = 150: 450 rows, the latency-organised forms."""
    N, D, T, k, al, be, seed = 3000, 100, 100, 20, [0.1, 0.1], [0.5, 0.95], 91
    nbt = 5
    E = nbt * bs
    src, dst, neg, ts, eidx = I.make_stream("general", N, E, seed)
    w = I.model_weights(D, F, T, len(al), seed)
    _, efeat = I.random_tables(N, E + 1, D, F, seed)
    dev = torch.device("cuda")
    t = [torch.from_numpy(x).to(dev) for x in (src, dst, neg, ts, eidx)]
    outs = {}
    for mode in ("seq", "pipe"):
        tgn = build_tgn(N, E + 1, D, F, T, k, al, be, w, efeat).eval()
        if mode == "pipe":
            tgn.enable_pipeline(tppr_cus=0, max_batch=bs)
        embs = []
        main = getattr(tgn, "main_stream", None) or torch.cuda.current_stream()
        with torch.cuda.stream(main):
            for b in range(nbt):
                cur = tuple(x[b * bs:(b + 1) * bs] for x in t)
                nxt = tuple(x[(b + 1) * bs:(b + 2) * bs] for x in t) if (mode == "pipe" and b + 1 < nbt) else None
                embs.append(tgn.step_device(*cur, prefetch=nxt).clone())
        torch.cuda.synchronize()
        outs[mode] = (torch.stack(embs).cpu().numpy(), tgn.memory.memory.cpu().numpy(), tgn.memory.last_update.cpu().numpy())
        tgn.enable_pipeline(False)
    for q in range(3):
        assert np.array_equal(outs["seq"][q], outs["pipe"][q])
    assert np.abs(outs["seq"][0]).max() > 0


@pytest.mark.parametrize("F,bs", [(1, 400), (1, 150)])
def test_out_gru_gate_survives_what_happens_between_fused_steps(F, bs):
    """The gate between the two halves of k_out_gru / k_out_gru2 keeps ALL its state in two words of the GRU workspace
    (count of source-path units, count of participants that have left; the last one out zeroes both): nothing the host
    does between two fused steps can leave it out of step with the kernels.  Driven through the sequences the round-5 review
    named -- the workspace re-packed mid-pipeline (a weight changed in place: zt_pipeline_update(weights_changed = 1)), a
    step whose GRU half has no rows (positions = (0, 0)) between fused steps, the pipeline destroyed and re-created on the
    same workspace, a direct zt_gru_update on that workspace between two pipeline steps -- against the sequential path
    (separate kernels), bit for bit; and the two gate words read zero after every step."""
    N, D, T, k, al, be, seed = 3000, 100, 100, 20, [0.1, 0.1], [0.5, 0.95], 97
    nbt = 9
    E = nbt * bs
    src, dst, neg, ts, eidx = I.make_stream("general", N, E, seed)
    w = I.model_weights(D, F, T, len(al), seed)
    _, efeat = I.random_tables(N, E + 1, D, F, seed)
    dev = torch.device("cuda")
    t = [torch.from_numpy(x).to(dev) for x in (src, dst, neg, ts, eidx)]
    outs = {}
    for mode in ("seq", "pipe"):
        tgn = build_tgn(N, E + 1, D, F, T, k, al, be, w, efeat).eval()
        if mode == "pipe":
            tgn.enable_pipeline(tppr_cus=0, max_batch=bs)
        embs, gates = [], []
        for b in range(nbt):
            cur = tuple(x[b * bs:(b + 1) * bs] for x in t)
            # (no batch is queried ahead: the pipeline may be torn down between any two steps without a T-PPR update
            #  having been applied for a batch that is then stepped again)
            main = getattr(tgn, "main_stream", None) or torch.cuda.current_stream()
            with torch.cuda.stream(main):
                if b == 2:                                   # a weight changes in place: the workspace is re-packed
                    with torch.no_grad():
                        tgn.memory_updater.memory_updater.weight_ih.mul_(1.0)
                        tgn.embedding_module.fc2.weight.mul_(1.0)
                if b == 6:                                   # a direct zt_gru_update on the pipeline's workspace (no row is flagged: the
                    ids = torch.arange(1, 2 * bs + 1, dtype=torch.int32, device=dev)     # tables stay, the row list and counter do not)
                    tgn.memory_updater.update_device(tgn.memory, ids, 2 * bs)
                pos = (0, 0) if b == 4 else None             # this step's GRU half finds no row
                embs.append(tgn.step_device(*cur, positions=pos).clone())
            torch.cuda.synchronize()
            if mode == "pipe":
                gates.append(tgn.memory_updater._ws[:256].view(torch.int32)[32:34].cpu().numpy().copy())
                if b == 5:                                   # the pipeline goes and comes back on the same workspace
                    ws_before = tgn.memory_updater._ws.data_ptr()
                    tgn.enable_pipeline(False)
                    tgn.enable_pipeline(tppr_cus=0, max_batch=bs)
                    assert tgn.memory_updater._ws.data_ptr() == ws_before
        torch.cuda.synchronize()
        tgn.embedding_module.tppr_finder.check_status()
        assert int(tgn.embedding_module._status.item()) == 0 if tgn.embedding_module._status is not None else True
        outs[mode] = (torch.stack(embs).cpu().numpy(), tgn.memory.memory.cpu().numpy(), tgn.memory.last_update.cpu().numpy())
        if mode == "pipe":
            assert all((g == 0).all() for g in gates), gates
        tgn.enable_pipeline(False)
    for q in range(3):
        assert np.array_equal(outs["seq"][q], outs["pipe"][q])
    assert np.abs(outs["seq"][0]).max() > 0


def test_out_gru_gate_gives_up_and_reports():
    """The gate's wait is bounded (4 s of the wall clock, as every wait of k_stream): with the count of source-path units
    poisoned so that it can never reach its target, the GRU half gives up, leaves the memory rows of its tiles untouched,
    writes ZT_ERR_TIMEOUT to the step's status word and to the pipeline's host-mapped latch -- the NEXT step call fails
    with it, once -- and the last participant out still zeroes the gate: the step after that runs clean."""
    from zebra_amd import _capi
    N, D, F, T, k, al, be, seed, bs = 3000, 100, 1, 100, 20, [0.1, 0.1], [0.5, 0.95], 98, 400
    E = 4 * bs
    src, dst, neg, ts, eidx = I.make_stream("general", N, E, seed)
    w = I.model_weights(D, F, T, len(al), seed)
    _, efeat = I.random_tables(N, E + 1, D, F, seed)
    dev = torch.device("cuda")
    t = [torch.from_numpy(x).to(dev) for x in (src, dst, neg, ts, eidx)]
    tgn = build_tgn(N, E + 1, D, F, T, k, al, be, w, efeat).eval()
    tgn.enable_pipeline(tppr_cus=0, max_batch=bs)
    batch = lambda b: tuple(x[b * bs:(b + 1) * bs] for x in t)
    with torch.cuda.stream(tgn.main_stream):
        tgn.step_device(*batch(0))
        torch.cuda.synchronize()
        gate = tgn.memory_updater._ws[:256].view(torch.int32)
        assert gate[32:34].tolist() == [0, 0]
        mem_before = tgn.memory.memory.clone()
        gate[32] = -3                                        # three units short for ever: every GRU workgroup's wait gives up
        torch.cuda.synchronize()
        tgn.step_device(*batch(1))                           # (enqueued: the failure is the kernel's, seconds from now)
        torch.cuda.synchronize()
        assert int(tgn.embedding_module._status.item()) == _capi.ZT_ERR_TIMEOUT
        assert torch.equal(tgn.memory.memory, mem_before)    # no tile wrote its rows
        assert gate[32:34].tolist() == [0, 0]                # the last one out reset the gate all the same
        with pytest.raises(_capi.ZebraError, match="status -5"):
            tgn.step_device(*batch(2))
        tgn.embedding_module._status.zero_()
        tgn.step_device(*batch(2))                           # reported once; the pipeline runs on
        tgn.step_device(*batch(3), check_status=True)
        torch.cuda.synchronize()
        assert gate[32:34].tolist() == [0, 0]
        assert not torch.equal(tgn.memory.memory, mem_before)
    tgn.enable_pipeline(False)


@pytest.mark.parametrize("group,look,ragged,release", [(2, 5, False, 0), (3, 8, True, 0), (4, 11, False, 0), (2, 2, False, 0),
                                                       (3, 1, True, 0), (8, 25, True, 0), (3, 8, True, _capi.RELEASE_LAUNCH),
                                                       (4, 11, False, _capi.RELEASE_LAUNCH)])
def test_grouped_tppr_launches_match_sequential(group, look, ragged, release):
    """zt_pipeline_set_group: the streaming T-PPR update of `group` consecutive batches as ONE launch (edges in
    order across the batches, every batch's rows in its own block) must not change any result -- whole batches and
    row shards, a full view ahead (3 * group - 1 batches), a short one (smaller groups come out), and a stream
    whose last batch is shorter.  release: how the aggregation of a batch learns that its rows are written -- the
    library's pick (round 6: a counter per batch inside the launch, the main stream passes a gate per batch) or the
    event behind the whole launch (ZT_CHOICE_GROUP_RELEASE)."""
    _capi.set_kernel_choice(_capi.CHOICE_GROUP_RELEASE, release)
    try:
        _grouped_launches_match_sequential(group, look, ragged)
    finally:
        _capi.set_kernel_choice(_capi.CHOICE_GROUP_RELEASE, 0)


def _grouped_launches_match_sequential(group, look, ragged):
    name = "d100_f1"
    N, E, D, F, T, k, al, be, seed, bs, nb = I.EMBED_CASES[name]
    src, dst, neg, ts, eidx = I.make_stream("general", N, E, seed)
    w = I.model_weights(D, F, T, len(al), seed)
    _, efeat = I.random_tables(N, E + 1, D, F, seed)
    dev = torch.device("cuda")
    t = [torch.from_numpy(x).to(dev) for x in (src, dst, neg, ts, eidx)]
    n_e = E - (bs // 3 if ragged else 0)
    cuts = list(range(0, n_e, bs)) + [n_e]
    batches = [tuple(x[a:b] for x in t) for a, b in zip(cuts[:-1], cuts[1:])]
    outs = {}
    for mode in ("seq", "grouped", "grouped_rows"):
        tgn = build_tgn(N, E + 1, D, F, T, k, al, be, w, efeat).eval()
        if mode != "seq":
            tgn.enable_pipeline(tppr_cus=0, max_batch=64, group=group)
        embs = []
        main = getattr(tgn, "main_stream", None) or torch.cuda.current_stream()
        with torch.cuda.stream(main):
            for b, cur in enumerate(batches):
                B = cur[0].numel()
                rows = (B // 2, 3 * B - 5) if mode == "grouped_rows" else None
                ahead = batches[b + 1: b + 1 + look] if mode != "seq" else None
                embs.append(tgn.step_device(*cur, rows=rows, ahead=ahead).clone())
        torch.cuda.synchronize()
        tgn.embedding_module.tppr_finder.check_status()
        outs[mode] = ([e.cpu().numpy() for e in embs], tgn.memory.memory.cpu().numpy(),
                      tgn.embedding_module.tppr_finder.export_state(0), tgn.embedding_module.tppr_finder.export_state(1))
        tgn.enable_pipeline(False)
    for b, cur in enumerate(batches):
        B = cur[0].numel()
        assert np.array_equal(outs["seq"][0][b], outs["grouped"][0][b]), "batch %d" % b
        assert np.array_equal(outs["seq"][0][b][B // 2: 3 * B - 5], outs["grouped_rows"][0][b]), "batch %d (rows)" % b
    for mode in ("grouped", "grouped_rows"):
        assert np.array_equal(outs["seq"][1], outs[mode][1])
        for m in (2, 3):
            for kk in outs["seq"][m]:
                assert np.array_equal(outs["seq"][m][kk], outs[mode][m][kk])


@pytest.mark.parametrize("F,n", [(172, 400), (172, 37), (1, 512), (1, 2000), (4, 1), (1, 8192), (172, 1203)])
def test_gru_kernels_agree(F, n):
    """The two organisations of the memory update -- k_gru (16 rows per workgroup, weights streamed per tile: the library's
    pick beyond 512 rows) and k_gru_split (a workgroup per 16 rows x N-tile, new rows committed by the tile's last workgroup)
    -- pinned one after the other (zt_set_kernel_choice) against torch's GRUCell on the same flagged rows: memory,
    last_update, the projected table; flags cleared; ragged last tile; rows the update does not name stay untouched."""
    from zebra_amd import _capi
    D = T = 100
    N, E1 = 12000, 100
    w = I.model_weights(D, F, T, 2, 31)
    _, efeat = I.random_tables(N, E1, D, F, 31)
    g = torch.Generator().manual_seed(F + n)
    msg = torch.randn((N, 2 * D + F + T), generator=g)
    mem0 = torch.randn((N, D), generator=g) * 0.3
    ts = torch.rand(N, generator=g) * 1e6
    ids = (torch.randperm(N - 1, generator=g)[:n] + 1).to(torch.int32)
    outs = {}
    tgn = build_tgn(N, E1, D, F, T, 20, [0.1, 0.1], [0.5, 0.95], w, efeat).eval()
    m = tgn.memory
    try:
        for mode, choice in (("tile", _capi.GRU_TILE), ("split", _capi.GRU_SPLIT), ("auto", 0)):
            _capi.set_kernel_choice(_capi.CHOICE_GRU, choice)
            m.messages.copy_(msg.cuda()); m.memory.copy_(mem0.cuda()); m.timestamps.copy_(ts.cuda())
            m.last_update.zero_()
            ids_d = ids.cuda()
            m._flag_buf[ids_d.long()] = 1
            table = tgn.embedding_module._projection(m)                     # the projected table follows the update
            tgn.memory_updater.update_device(m, ids_d, ids_d.numel())
            torch.cuda.synchronize()
            outs[mode] = dict(mem=m.memory.cpu().numpy(), lu=m.last_update.cpu().numpy(), flags=m._flag_buf.cpu().numpy()[:N])
    finally:
        _capi.set_kernel_choice(_capi.CHOICE_GRU, 0)
    cell = torch.nn.GRUCell(2 * D + F + T, D)
    with torch.no_grad():
        cell.weight_ih.copy_(torch.from_numpy(w["w_ih"])); cell.weight_hh.copy_(torch.from_numpy(w["w_hh"]))
        cell.bias_ih.copy_(torch.from_numpy(w["b_ih"])); cell.bias_hh.copy_(torch.from_numpy(w["b_hh"]))
        want = mem0.clone()
        want[ids.long()] = cell(msg[ids.long()], mem0[ids.long()])
    for mode, o in outs.items():
        assert np.abs(o["mem"] - want.numpy()).max() <= 1e-5, mode
        lu = np.zeros(N, np.float32); lu[ids.numpy()] = ts.numpy()[ids.numpy()]
        assert np.array_equal(o["lu"], lu), mode
        assert not o["flags"].any(), mode
    assert np.abs(outs["split"]["mem"] - outs["tile"]["mem"]).max() <= 2e-6


@pytest.mark.parametrize("F,B", [(1, 4096), (172, 600), (300, 50), (4, 7)])
def test_message_kernels_agree(oracle, F, B):
    """k_build_messages2 (two batch positions per wavefront: the library's pick for D, T <= 128, F <= 256) against
    k_build_messages (one; pinned with zt_set_kernel_choice; F = 300 takes it either way) on the same batch -- self-loops, a
    node at many positions, a position shard -- : messages, timestamps, the winners' list and the flags equal bit for bit,
    the scratch back at -1; and against the oracle's last-message rule (model/tgn_model.py:204-226).  An out-of-range id
    rejects the whole call (ZT_ERR_RANGE) and leaves the tables and the scratch as they were, with either kernel."""
    from zebra_amd import _capi
    D = T = 100
    N, E1 = 3000, 9000
    w = I.model_weights(D, F, T, 2, 5)
    mem_t, efeat = I.random_tables(N, E1, D, F, 5)
    rng = np.random.RandomState(F + B)
    src = rng.randint(1, N, B).astype(np.int32); dst = rng.randint(1, N, B).astype(np.int32)
    src[::7] = 11                                    # a node at many positions
    dst[3::11] = src[3::11]                          # self-loops
    ts = np.cumsum(rng.rand(B) * 50.0) + 1.0e5
    eidx = rng.randint(1, E1, B).astype(np.int64)
    tgn = build_tgn(N, E1, D, F, T, 20, [0.1, 0.1], [0.5, 0.95], w, efeat).eval()
    m = tgn.memory
    lu0 = (torch.rand(N) * 9.0e4)
    outs = {}
    try:
        for mode, choice in (("two", 0), ("one", _capi.MSG_ONE)):
            _capi.set_kernel_choice(_capi.CHOICE_MESSAGES, choice)
            for pos in (None, (B // 3, 2 * B - B // 5)):
                m.memory.copy_(torch.from_numpy(mem_t).cuda()); m.last_update.copy_(lu0.cuda())
                m.messages.zero_(); m.timestamps.zero_(); m._flag_buf.zero_()
                args = [torch.from_numpy(a).cuda() for a in (src, dst, ts, eidx)]
                tgn.store_messages_device(*args, pos_range=pos)
                torch.cuda.synchronize()
                assert int(tgn._status.item()) == 0
                outs[(mode, pos is None)] = dict(msg=m.messages.cpu().numpy(), ts=m.timestamps.cpu().numpy(),
                                                 flags=m._flag_buf.cpu().numpy()[:N].copy(), scratch=tgn._scratch.cpu().numpy())
            # an id out of range: the call is rejected as a whole
            bad = src.copy(); bad[B // 2] = N + 5
            before = m.messages.clone()
            tgn.store_messages_device(torch.from_numpy(bad).cuda(), *args[1:])
            torch.cuda.synchronize()
            assert int(tgn._status.item()) == _capi.ZT_ERR_RANGE, mode
            tgn._status.zero_()
            assert torch.equal(before, m.messages) and bool((tgn._scratch == -1).all()), mode
    finally:
        _capi.set_kernel_choice(_capi.CHOICE_MESSAGES, 0)
    for whole in (True, False):
        a, b = outs[("two", whole)], outs[("one", whole)]
        for kk in a:
            assert np.array_equal(a[kk], b[kk]), (kk, whole)
        assert (a["scratch"] == -1).all()
    # the oracle's rule on the whole batch
    mo = oracle.MemoryOracle(N, D, 2 * D + F + T)
    mo.memory[:] = mem_t; mo.last_update[:] = lu0.numpy()
    mo.store_messages(efeat, I.time_encode_weights(T), src, dst, ts, eidx)
    got = outs[("two", True)]
    touched = np.unique(np.concatenate([src, dst]))
    assert np.abs(got["msg"][touched] - mo.messages[touched]).max() <= 1e-4
    assert np.array_equal(got["ts"][touched], mo.timestamps[touched])
    assert np.array_equal(got["flags"].astype(bool), mo.flags.astype(bool)[:N])


@pytest.mark.parametrize("F,k,n", [(1, 20, 12288), (1, 40, 3000), (172, 20, 1800), (172, 40, 1030), (1, 20, 5), (4, 20, 2051)])
def test_output_layer_kernels_agree(F, k, n):
    """The three organisations of the output layers (fc2 of every model on the reduced rows, the source transform;
    modules/embedding_module.py:243-246,272-276,320-328): k_embed_out (32-row workgroups, weights streamed), k_embed_out2
    (a wave per 16 rows x path x N-tile: the pick up to 1 024 rows) and k_embed_out3 (persistent, weights resident in LDS,
    a wave per 16 rows x all N-tiles: beyond), pinned one after the other on the same aggregation: equal to the bit (the
    same sums in the same order), a ragged last tile, the partial-sum groups of the wide aggregation kernel included."""
    from zebra_amd import _capi
    D = T = 100
    N, E1 = 5000, 20000
    g = torch.Generator().manual_seed(3 + F + k + n)
    w = I.model_weights(D, F, T, 2, 9)
    _, efeat = I.random_tables(N, E1, D, F, 9)
    tgn = build_tgn(N, E1, D, F, T, k, [0.1, 0.1], [0.5, 0.95], w, efeat).eval()
    dev = tgn.device
    tgn.memory.memory.copy_(torch.randn((N, D), generator=g).to(dev))
    nodes = torch.randint(0, N, (n,), generator=g, dtype=torch.int32).to(dev)
    on = torch.randint(0, N, (2, n, k), generator=g, dtype=torch.int32)
    oe = torch.randint(0, E1, (2, n, k), generator=g, dtype=torch.int32)
    od = torch.rand((2, n, k), generator=g) * 3.0e6
    ow = torch.rand((2, n, k), generator=g)
    ow[:, ::5] = 0.0                                      # rows whose weights sum to 0 (the bias of fc2 drops out: S = 0)
    args = [t.to(dev).contiguous() for t in (on, oe, od.float(), ow.float())]
    em = tgn.embedding_module
    outs = {}
    try:
        for mode, choice in (("tiled", _capi.OUT_TILED), ("latency", _capi.OUT_LATENCY), ("persist", _capi.OUT_PERSIST), ("auto", 0)):
            _capi.set_kernel_choice(_capi.CHOICE_EMBED_OUT, choice)
            outs[mode] = em.embed_device(tgn.memory.memory, nodes, *args, memory_obj=tgn.memory).cpu().numpy()
    finally:
        _capi.set_kernel_choice(_capi.CHOICE_EMBED_OUT, 0)
    for mode in ("latency", "persist", "auto"):
        assert np.array_equal(outs["tiled"], outs[mode]), mode


def test_pipeline_with_wide_k_matches_sequential():
    """k = 64 through the native step: the wide streaming path takes one batch per T-PPR launch (a requested group of 4 is
    clamped), runs ahead on its stream like any other, and the step's results equal the sequential protocol's bit for bit."""
    N, E, D, F, T, k, al, be, seed, bs = 200, 1200, 100, 1, 100, 64, [0.1, 0.1], [0.5, 0.95], 77, 100
    src, dst, neg, ts, eidx = I.make_stream("general", N, E, seed)
    w = I.model_weights(D, F, T, len(al), seed)
    _, efeat = I.random_tables(N, E + 1, D, F, seed)
    dev = torch.device("cuda")
    t = [torch.from_numpy(x).to(dev) for x in (src, dst, neg, ts, eidx)]
    batches = [tuple(x[a:a + bs] for x in t) for a in range(0, E, bs)]
    outs = {}
    for mode in ("seq", "pipe"):
        tgn = build_tgn(N, E + 1, D, F, T, k, al, be, w, efeat).eval()
        if mode == "pipe":
            tgn.enable_pipeline(tppr_cus=0, max_batch=128, group=4)
        main = getattr(tgn, "main_stream", None) or torch.cuda.current_stream()
        with torch.cuda.stream(main):
            embs = [tgn.step_device(*cur, ahead=batches[b + 1: b + 6] if mode == "pipe" else None).clone()
                    for b, cur in enumerate(batches)]
        torch.cuda.synchronize()
        tgn.embedding_module.tppr_finder.check_status()
        outs[mode] = (torch.stack(embs).cpu().numpy(), tgn.memory.memory.cpu().numpy(),
                      [tgn.embedding_module.tppr_finder.export_state(m) for m in range(2)])
        tgn.enable_pipeline(False)
    assert np.array_equal(outs["seq"][0], outs["pipe"][0])
    assert np.array_equal(outs["seq"][1], outs["pipe"][1])
    for a, b in zip(outs["seq"][2], outs["pipe"][2]):
        for kk in a:
            assert np.array_equal(a[kk], b[kk]), kk


def test_group_members_are_released_when_the_launch_is_rejected():
    """A launch group whose T-PPR update the prepass rejects (a node id out of range in ONE of its batches: ZT_ERR_RANGE, the
    state untouched, every output row an empty dictionary) must still let the main stream through: the rejected launch opens
    the gate of every member, no gate wait gives up (the pipeline's latch stays clear), the error is reported by the finder's
    status, and the steps after it run."""
    name = "d100_f1"
    N, E, D, F, T, k, al, be, seed, bs, nb = I.EMBED_CASES[name]
    src, dst, neg, ts, eidx = I.make_stream("general", N, E, seed)
    w = I.model_weights(D, F, T, len(al), seed)
    _, efeat = I.random_tables(N, E + 1, D, F, seed)
    dev = torch.device("cuda")
    tgn = build_tgn(N, E + 1, D, F, T, k, al, be, w, efeat).eval()
    tgn.enable_pipeline(tppr_cus=0, max_batch=64, group=3)
    neg_bad = neg.copy()
    neg_bad[2 * bs + 3] = N + 7                              # third batch: a negative sample that is no node
    t = [torch.from_numpy(x).to(dev) for x in (src, dst, neg_bad, ts, eidx)]
    batches = [tuple(x[a:a + bs] for x in t) for a in range(0, 7 * bs, bs)]
    t0 = time.perf_counter()
    embs, refused = [], False
    with torch.cuda.stream(tgn.main_stream):
        # batch 0 alone, batches 1 .. 3 as one launch (rejected), 4 .. 6 as the next.  The handle's host-mapped latch may
        # refuse a later launch as soon as the rejected one has run: either way nothing may wait for rows that never come
        try:
            for b, cur in enumerate(batches[:4]):
                embs.append(tgn.step_device(*cur, ahead=batches[b + 1: b + 9]).clone())
        except IndexError:
            refused = True
    torch.cuda.synchronize()
    assert time.perf_counter() - t0 < 3.0, "a gate waited for a launch that was rejected"
    if not refused:
        with pytest.raises(IndexError):
            tgn.embedding_module.tppr_finder.check_status()
    else:
        try:
            tgn.embedding_module.tppr_finder.check_status()      # (reports and clears what is left of it)
        except IndexError:
            pass
    assert all(torch.isfinite(e).all() for e in embs)      # (how many steps were taken before the latch was seen is a matter of timing)
    tgn.embedding_module._status.zero_()
    tgn.enable_pipeline(False)                                # (the groups staged around the refused call are void)
    tgn.enable_pipeline(tppr_cus=0, max_batch=64, group=3)
    with torch.cuda.stream(tgn.main_stream):                  # no latched time-out: the next steps are taken
        for b in range(4, 7):
            tgn.step_device(*batches[b], ahead=batches[b + 1: b + 9])
    torch.cuda.synchronize()
    tgn.embedding_module.tppr_finder.check_status()
    tgn.enable_pipeline(False)


@pytest.mark.parametrize("strategy,group", [("streaming", 1), ("streaming", 3), ("pruning", 1)])
def test_native_batch_loop_matches_stepwise(strategy, group):
    """zt_pipeline_run (TGN.run_device: n steps from one host call, the batch loop of evaluation/evaluation.py:19-45) against
    the same steps made one by one through step_device: every step's embeddings, the scores of the last step, and the
    state (T-PPR rows, memory tables) afterwards -- bit for bit; a ragged last batch included."""
    import types
    from zebra_amd.tppr import get_neighbor_finder
    D = T = 100
    N, E, F, bs, k, al, be, seed = 3000, 4070, 1, 400, 20, [0.1, 0.1], [0.5, 0.95], 77
    src, dst, neg, ts, eidx = I.make_stream("general", N, E, seed)
    w = I.model_weights(D, F, T, 2, seed)
    _, efeat = I.random_tables(N, E + 1, D, F, seed)
    nf = get_neighbor_finder(types.SimpleNamespace(sources=src, destinations=dst, edge_idxs=eidx, timestamps=ts)) \
        if strategy == "pruning" else None
    res = {}
    for mode in ("step", "run"):
        tgn = build_tgn(N, E + 1, D, F, T, k, al, be, w, efeat, strategy=strategy, nf=nf).eval()
        tgn.enable_pipeline(tppr_cus=0, max_batch=512, group=group)
        tgn.enable_scoring()
        dev = tgn.device
        t = [torch.from_numpy(x).to(dev) for x in (src, dst, neg, ts, eidx)]
        batches = [tuple(x[s0:s0 + bs] for x in t) for s0 in range(0, E, bs)]            # the last one has 70 edges
        full = [b for b in batches if b[0].numel() == bs]
        with torch.cuda.stream(tgn.main_stream):
            if mode == "step":
                embs = [tgn.step_device(*cur, ahead=batches[q + 1: q + 1 + 3 * group]).clone() for q, cur in enumerate(batches)]
                out = torch.stack(embs[:len(full)])
                last = embs[-1]
            else:
                out = torch.empty((len(full), 3 * bs, 300), dtype=torch.float32, device=dev)
                tgn.run_device(tgn.prepare_run(full), out=out)
                last = tgn.step_device(*batches[-1])
            prob = tgn.last_prob().clone()
        torch.cuda.synchronize()
        st = [tgn.embedding_module.tppr_finder.export_state(m) for m in range(2)] if strategy == "streaming" else []
        res[mode] = (out, last, prob, tgn.memory.memory.clone(), tgn.memory.messages.clone(), tgn.memory.last_update.clone(), st)
        tgn.enable_pipeline(False)
    a, b = res["step"], res["run"]
    for q in range(6):
        assert torch.equal(a[q], b[q]), q
    for x, y in zip(a[6], b[6]):
        for kk in x:
            assert np.array_equal(x[kk], y[kk]), kk


def test_pipeline_batch_size_change_keeps_packed_weights():
    """A pipelined stream whose LAST batch is shorter by enough to move round_up(4 * 2B, 256) -- bs = 64 then 20 --
    (round-2 advisor finding: the GRU's packed weights used to sit behind the row list of the workspace, whose
    length follows the batch; the shorter batch then read them from the wrong place).  Results must equal the
    sequential path bit for bit: embeddings of every batch, the whole memory, pending messages."""
    N, D, F, T, k, al, be, seed, bs = 300, 100, 1, 100, 20, [0.1, 0.1], [0.5, 0.95], 57, 64
    E = 5 * bs + 20
    src, dst, neg, ts, eidx = I.make_stream("general", N, E, seed)
    w = I.model_weights(D, F, T, len(al), seed)
    _, efeat = I.random_tables(N, E + 1, D, F, seed)
    dev = torch.device("cuda")
    t = [torch.from_numpy(x).to(dev) for x in (src, dst, neg, ts, eidx)]
    cuts = list(range(0, E, bs)) + [E]
    batches = [tuple(x[a:b] for x in t) for a, b in zip(cuts[:-1], cuts[1:])]
    assert batches[-1][0].numel() == 20
    outs = {}
    for mode in ("seq", "pipe"):
        tgn = build_tgn(N, E + 1, D, F, T, k, al, be, w, efeat).eval()
        if mode == "pipe":
            tgn.enable_pipeline(tppr_cus=0, max_batch=64, group=2)
        embs = []
        main = getattr(tgn, "main_stream", None) or torch.cuda.current_stream()
        with torch.cuda.stream(main):
            for b, cur in enumerate(batches):
                embs.append(tgn.step_device(*cur, ahead=batches[b + 1: b + 6] if mode == "pipe" else None).clone())
        torch.cuda.synchronize()
        tgn.embedding_module.tppr_finder.check_status()
        outs[mode] = ([e.cpu().numpy() for e in embs], tgn.memory.memory.cpu().numpy(), tgn.memory.messages.cpu().numpy(),
                      tgn.memory.last_update.cpu().numpy())
        tgn.enable_pipeline(False)
    for b in range(len(batches)):
        assert np.array_equal(outs["seq"][0][b], outs["pipe"][0][b]), "batch %d" % b
    for q in (1, 2, 3):
        assert np.array_equal(outs["seq"][q], outs["pipe"][q])


def test_pipeline_refuses_to_drop_an_applied_group():
    """Leaving the announced order: three groups whose T-PPR update has been launched are waiting, the caller
    presents a batch none of them holds.  The streaming state already contains those edges -- dropping a group
    would apply them twice later -- so the step must fail instead (round-2 advisor finding)."""
    N, D, F, T, k, al, be, seed, bs = 300, 100, 1, 100, 20, [0.1, 0.1], [0.5, 0.95], 58, 16
    E = 12 * bs
    src, dst, neg, ts, eidx = I.make_stream("general", N, E, seed)
    w = I.model_weights(D, F, T, len(al), seed)
    _, efeat = I.random_tables(N, E + 1, D, F, seed)
    dev = torch.device("cuda")
    t = [torch.from_numpy(x).to(dev) for x in (src, dst, neg, ts, eidx)]
    batches = [tuple(x[b * bs:(b + 1) * bs] for x in t) for b in range(E // bs)]
    tgn = build_tgn(N, E + 1, D, F, T, k, al, be, w, efeat).eval()
    tgn.enable_pipeline(tppr_cus=0, max_batch=64, group=1)
    with torch.cuda.stream(tgn.main_stream):
        tgn.step_device(*batches[0], ahead=batches[1:3])          # 1 is queried ahead, 2 planned
        assert tgn.pipeline_outstanding() == 1
        # out of order: 5 is queried now (the free slot), 6 ahead (the slot of the merely PLANNED batch 2 may be
        # dropped), and then nothing is left for 7: the slots hold 5, 1 and 6, all applied to the state already
        with pytest.raises(ValueError):
            tgn.step_device(*batches[5], ahead=batches[6:8])
    torch.cuda.synchronize()
    tgn.enable_pipeline(False)


@pytest.mark.parametrize("perm", [None, 9])
def test_bench_path_vs_oracle(oracle, perm):
    """What bench.py times -- TGN.step_device through the native pipeline with CU-masked streams (T-PPR on 64 CUs),
    grouped T-PPR launches (two batches per launch), hub chains active, the specialised aggregation kernel and the
    projected memory table, at bench.py's batch size -- against the CPU oracle's protocol DIRECTLY (round-2 review:
    this path was only ever compared with the sequential HIP path).  Power-law stream over 100 K nodes, bs = 4096,
    a short fill, then 12 checked batches: embeddings of every batch <= 1e-4, then the T-PPR state bit-exact and
    memory / last_update / messages / flags."""
    from zebra_amd import synth
    N, D, F, T, k, bs = 100_000, 100, 1, 100, 20, 4096
    al, be = [0.1, 0.1], [0.5, 0.95]
    fill, nb = 6, 12
    E = (fill + nb) * bs
    src, dst, ts, eidx = synth.power_law_stream(N, E, seed=301, perm_seed=perm)
    neg = synth.negatives(dst, E, seed=302)
    w = I.model_weights(D, F, T, len(al), 303)
    efeat = synth.edge_features(E + 1, F, seed=304)
    tw = I.time_encode_weights(T)
    dev = torch.device("cuda")
    t = [torch.from_numpy(x).to(dev) for x in (src, dst, neg, ts, eidx)]
    batches = [tuple(x[b * bs:(b + 1) * bs] for x in t) for b in range(fill + nb)]
    tgn = build_tgn(N + 1, E + 1, D, F, T, k, al, be, w, efeat).eval()
    tgn.enable_pipeline(tppr_cus=64, group=2)
    p = oracle.ProtocolOracle(N + 1, D, F, T, k, al, be, w, efeat, tw, n_threads=8)
    embs = []
    with torch.cuda.stream(tgn.main_stream):
        for b, cur in enumerate(batches):
            e = tgn.step_device(*cur, ahead=batches[b + 1: b + 7])
            if b >= fill:
                embs.append(e.clone())
    torch.cuda.synchronize()
    tgn.embedding_module.tppr_finder.check_status()
    worst = 0.0
    for b in range(fill + nb):
        s, e = b * bs, (b + 1) * bs
        ref, _ = p.batch(src[s:e], dst[s:e], neg[s:e], ts[s:e], eidx[s:e], False)
        if b >= fill:
            worst = max(worst, float(np.abs(embs[b - fill].cpu().numpy() - ref).max()))
    assert worst <= TOL, "embeddings differ from the oracle by %g" % worst
    f = tgn.embedding_module.tppr_finder
    for m in range(len(al)):
        a, bb = f.export_state(m), p.tppr.export(m)
        for kk in a:
            assert np.array_equal(a[kk], bb[kk]), "T-PPR state %s of model %d differs from the oracle" % (kk, m)
    assert np.abs(tgn.memory.memory.cpu().numpy() - p.mem.memory).max() <= TOL
    assert np.array_equal(tgn.memory.last_update.cpu().numpy(), p.mem.last_update)
    assert np.abs(tgn.memory.messages.cpu().numpy() - p.mem.messages).max() <= TOL
    assert np.array_equal(tgn.memory.nodes.astype(np.uint8), p.mem.flags)
    tgn.enable_pipeline(False)


def test_reference_surface_reaches_the_pipeline(oracle):
    """compute_temporal_embeddings (numpy batches, the reference's surface) runs through the native pipeline once it
    is enabled -- average_topk then comes from the slot's weights (zt_pipeline_set_stats) -- with the oracle's
    results."""
    name = "d100_f1"
    N, E, D, F, T, k, al, be, seed, bs, nb = I.EMBED_CASES[name]
    src, dst, neg, ts, eidx = I.make_stream("general", N, E, seed)
    w = I.model_weights(D, F, T, len(al), seed)
    _, efeat = I.random_tables(N, E + 1, D, F, seed)
    tgn = build_tgn(N, E + 1, D, F, T, k, al, be, w, efeat).eval()
    tgn.enable_pipeline(tppr_cus=0, max_batch=64)
    p = oracle.ProtocolOracle(N, D, F, T, k, al, be, w, efeat, I.time_encode_weights(T))
    for b in range(E // bs):
        s, e = b * bs, (b + 1) * bs
        with torch.no_grad():
            se, de, ne = tgn.compute_temporal_embeddings(src[s:e], dst[s:e], neg[s:e], ts[s:e], eidx[s:e], 10, False)
        ref, _ = p.batch(src[s:e], dst[s:e], neg[s:e], ts[s:e], eidx[s:e], False)
        assert np.abs(torch.cat([se, de, ne]).cpu().numpy() - ref).max() <= TOL
        assert abs(tgn.embedding_module.average_topk - p.average_topk) < 1e-6
    assert tgn.pipeline_outstanding() == 0
    tgn.enable_pipeline(False)


def test_pipeline_follows_backup_and_restore():
    """The native pipeline holds raw pointers to the memory tables, the T-PPR handle and the workspaces: replacing
    any of them between steps (restore_memory / restore_tppr of the epoch protocol, train.py:296-306, a call
    outside the pipeline that grows a workspace) must be noticed -- same results as without the pipeline."""
    name = "d100_f1"
    N, E, D, F, T, k, al, be, seed, bs, nb = I.EMBED_CASES[name]
    src, dst, neg, ts, eidx = I.make_stream("general", N, E, seed)
    w = I.model_weights(D, F, T, len(al), seed)
    _, efeat = I.random_tables(N, E + 1, D, F, seed)
    dev = torch.device("cuda")
    t = [torch.from_numpy(x).to(dev) for x in (src, dst, neg, ts, eidx)]
    batches = [tuple(x[b * bs:(b + 1) * bs] for x in t) for b in range(E // bs)]
    outs = {}
    for pipe in (False, True):
        tgn = build_tgn(N, E + 1, D, F, T, k, al, be, w, efeat).eval()
        if pipe:
            tgn.enable_pipeline(tppr_cus=0, max_batch=64, group=2)
        em = tgn.embedding_module
        main = getattr(tgn, "main_stream", None) or torch.cuda.current_stream()
        embs = []

        def steps(lo, hi):
            with torch.cuda.stream(main):
                for b in range(lo, hi):
                    ahead = batches[b + 1: min(hi, b + 6)] if pipe else None
                    embs.append(tgn.step_device(*batches[b], ahead=ahead).clone())
            torch.cuda.synchronize()

        steps(0, 4)
        mem_bak = tgn.memory.backup_memory()
        tppr_bak = em.backup_tppr()
        steps(4, 8)
        tgn.memory.restore_memory(mem_bak)              # new tensors behind the same names
        em.restore_tppr(tppr_bak)
        tgn.memory_updater.update_memory_in_test(tgn.memory)      # a call outside the pipeline (grows the GRU workspace)
        steps(4, 9)
        outs[pipe] = (torch.stack(embs).cpu().numpy(), tgn.memory.memory.cpu().numpy(), em.tppr_finder.export_state(0))
        if pipe:
            tgn.enable_pipeline(False)
    assert np.array_equal(outs[False][0], outs[True][0])
    assert np.array_equal(outs[False][1], outs[True][1])
    for kk in outs[False][2]:
        assert np.array_equal(outs[False][2][kk], outs[True][2][kk])


def test_full_size_step_properties():
    """The whole eval step at BASELINE.json's full-size configuration (C5: 10 M nodes, bs=4096, k=20,
    D=T=100, F=1), checked through size-independent properties: (1) the pipelined schedule (T-PPR one
    batch ahead on a CU-masked stream, prepass two batches ahead) equals the sequential one bit for bit
    -- embeddings, touched memory rows, pending messages; (2) rows of a batch are independent given
    the pre-batch state (modules/embedding_module.py:243-276): embedding a sub-range of the rows gives
    the same values as the full call; (3) only the batch's endpoints change in the memory."""
    from zebra_amd import synth
    wl = synth.WORKLOADS["c5"]
    N, B, k, D, F, T = wl["n_nodes"], 4096, 20, 100, 1, 100
    al, be = [0.1, 0.1], [0.5, 0.95]
    nb = 12
    src, dst, ts, eidx = synth.power_law_stream(N, nb * B, seed=91)
    neg = synth.negatives(dst, len(src), seed=92)
    w = I.model_weights(D, F, T, len(al), 93)
    efeat = synth.edge_features(nb * B + 1, F, seed=94)
    dev = torch.device("cuda")
    t = [torch.from_numpy(x).to(dev) for x in (src, dst, neg, ts, eidx)]
    touched = torch.from_numpy(np.unique(np.concatenate([src, dst])).astype(np.int64)).to(dev)
    res = {}
    for mode in ("seq", "pipe"):
        tgn = build_tgn(N + 1, nb * B + 1, D, F, T, k, al, be, w, efeat).eval()
        g = torch.Generator(device="cuda").manual_seed(95)
        tgn.memory.memory.copy_(torch.randn(tgn.memory.memory.shape, generator=g, device="cuda") * 0.1)
        if mode == "pipe":
            tgn.enable_pipeline(tppr_cus=48)
        main = getattr(tgn, "main_stream", None) or torch.cuda.current_stream()
        before = tgn.memory.memory.clone() if mode == "seq" else None
        embs = []
        with torch.cuda.stream(main):
            for b in range(nb):
                cur = tuple(x[b * B:(b + 1) * B] for x in t)
                nxt = tuple(x[(b + 1) * B:(b + 2) * B] for x in t) if (mode == "pipe" and b + 1 < nb) else None
                nx2 = tuple(x[(b + 2) * B:(b + 3) * B] for x in t) if (mode == "pipe" and b + 2 < nb) else None
                embs.append(tgn.step_device(*cur, prefetch=nxt, plan=nx2).clone())
        torch.cuda.synchronize()
        tgn.embedding_module.tppr_finder.check_status()
        res[mode] = (torch.stack(embs), tgn.memory.memory.index_select(0, touched),
                     tgn.memory.messages.index_select(0, touched), tgn.memory.last_update.index_select(0, touched))
        if mode == "seq":
            # (3) untouched rows keep their values
            mask = torch.ones(N + 1, dtype=torch.bool, device=dev)
            mask[touched] = False
            assert torch.equal(tgn.memory.memory[mask], before[mask])
            # (2) a sub-range of rows, embedded on its own from the final state
            em = tgn.embedding_module
            b = nb - 1
            nodes = torch.cat([t[0][b * B:(b + 1) * B], t[1][b * B:(b + 1) * B], t[2][b * B:(b + 1) * B]])
            on, oe, od, ow = em.tppr_finder.stream_device(nodes, t[3][b * B:(b + 1) * B], t[4][b * B:(b + 1) * B], 3, True, -1)
            full = em.embed_device(tgn.memory.memory, nodes, on, oe, od, ow)
            lo, hi = 1000, 1777
            part = em.embed_device(tgn.memory.memory, nodes[lo:hi].contiguous(), on[:, lo:hi].contiguous(),
                                   oe[:, lo:hi].contiguous(), od[:, lo:hi].contiguous(), ow[:, lo:hi].contiguous())
            assert torch.equal(full[lo:hi], part)
            assert torch.isfinite(full).all()
        tgn.enable_pipeline(False)
        del tgn
        torch.cuda.empty_cache()
    for a, b in zip(res["seq"], res["pipe"]):
        assert torch.equal(a, b)


def test_device_side_evaluation_matches_reference_protocol():
    """zebra_amd.evaluation.eval_edge_prediction (probabilities and AP / AUC / accuracy
    stay on the device) against the reference's protocol restated with scikit-learn on the host
    (evaluation/evaluation.py:7-48) -- same model, same sampler seed."""
    import math
    import types
    sk = pytest.importorskip("sklearn.metrics")
    from zebra_amd import evaluation as ev
    name = "d100_f1"
    N, E, D, F, T, k, al, be, seed, bs, nb = I.EMBED_CASES[name]
    src, dst, neg, ts, eidx = I.make_stream("general", N, E, seed)
    w = I.model_weights(D, F, T, len(al), seed)
    _, efeat = I.random_tables(N, E + 1, D, F, seed)
    data = types.SimpleNamespace(sources=src, destinations=dst, timestamps=ts, edge_idxs=eidx, n_interactions=len(src))

    class Sampler:                       # RandEdgeSampler's interface (utils/util.py:54-84)
        def __init__(self, dsts, seed):
            self.seed, self.dst_list = seed, np.unique(dsts)
            self.random_state = np.random.RandomState(seed)

        def reset_random_state(self):
            self.random_state = np.random.RandomState(self.seed)

        def sample(self, size):
            i = self.random_state.randint(0, len(self.dst_list), size)
            return self.dst_list[i], self.dst_list[self.random_state.randint(0, len(self.dst_list), size)]

    batch = 150
    got = ev.eval_edge_prediction(build_tgn(N, E + 1, D, F, T, k, al, be, w, efeat), Sampler(dst, 7), data, 10, batch)
    # the reference's loop, on the host
    model = build_tgn(N, E + 1, D, F, T, k, al, be, w, efeat).eval()
    smp = Sampler(dst, 7)
    smp.reset_random_state()
    aps, aucs, accs = [], [], []
    with torch.no_grad():
        for b in range(math.ceil(len(src) / batch)):
            s, e = b * batch, min(len(src), (b + 1) * batch)
            _, negs = smp.sample(e - s)
            pos, ng = model.compute_edge_probabilities(src[s:e], dst[s:e], negs, ts[s:e], eidx[s:e], 10, train=False)
            pos, ng = pos.cpu().numpy(), ng.cpu().numpy()
            y = np.concatenate([np.ones(e - s), np.zeros(e - s)])
            sc = np.concatenate([pos, ng])
            aps.append(sk.average_precision_score(y, sc))
            aucs.append(sk.roc_auc_score(y, sc))
            accs.append(sk.accuracy_score(np.zeros(e - s), np.argmax(np.hstack([pos, ng]), axis=1)))
    want = (np.mean(aps), np.mean(aucs), np.mean(accs))
    assert np.allclose(got, want, rtol=0, atol=1e-9), (got, want)


def test_training_step_gradients_match_reference():
    """SURVEY.md 8f-1 (first half): an unmodified training step in the reference's style
    (train.py:205-215: BCE on positive / negative probabilities, loss.backward()) through the drop-in
    gives the reference's loss and parameter gradients (fixture g8_train_grads, generated from the
    reference): T-PPR and memory kernels in HIP, the aggregation differentiable through torch device ops."""
    name = "d20_f7"
    N, E, D, F, T, k, al, be, seed, bs, nb = I.EMBED_CASES[name]
    g = golden("g8_train_grads")
    src, dst, neg, ts, eidx = I.make_stream("general", N, E, seed)
    w = I.model_weights(D, F, T, len(al), seed)
    _, efeat = I.random_tables(N, E + 1, D, F, seed)
    tgn = build_tgn(N, E + 1, D, F, T, k, al, be, w, efeat)
    tgn.train(True)
    crit = torch.nn.BCELoss()
    dev = torch.device("cuda")
    seen = 0
    for b in range(nb):
        s, e = b * bs, (b + 1) * bs
        tgn.zero_grad()
        pos, negp = tgn.compute_edge_probabilities(src[s:e], dst[s:e], neg[s:e], ts[s:e], eidx[s:e], 10, True)
        loss = crit(pos.squeeze(), torch.ones(bs, device=dev)) + crit(negp.squeeze(), torch.zeros(bs, device=dev))
        loss.backward()
        assert abs(float(loss.item()) - float(g["b%d_loss" % b])) <= 1e-5, "loss of batch %d" % b
        for pn, p in tgn.named_parameters():
            key = "b%d_grad_%s" % (b, pn)
            if key in g.files:
                assert p.grad is not None, pn
                want = g[key]
                err = np.abs(p.grad.detach().cpu().numpy() - want).max()
                assert err <= 1e-5 + 1e-4 * np.abs(want).max(), "%s in batch %d: %g" % (pn, b, err)
                seen += 1
        tgn.memory.detach_memory()
    assert seen >= 12 * nb


# ---------------------------------------------------------------------------------------------------------
# TimeEncode on the device (model/time_encoding.py:18-28), incl. the large-argument branch of time_cosf
# ---------------------------------------------------------------------------------------------------------
def _time_identity_tgn(N, E1, k=4):
    """A TGN whose aggregation returns cos(dt * w) itself: fc1 = [0 | 0 | I_T] with bias +1 (so the ReLU is
    the identity on cos + 1), fc2 = I with bias -1, one neighbour of weight 1 per row."""
    D = T = 100
    F = 1
    w = I.model_weights(D, F, T, 1, 5)
    w["fc1_w"] = np.concatenate([np.zeros((D, D + F), np.float32), np.eye(T, dtype=np.float32)], axis=1)
    w["fc1_b"] = np.ones(D, np.float32)
    w["fc2_w"] = np.eye(D, dtype=np.float32)
    w["fc2_b"] = -np.ones(D, np.float32)
    efeat = np.zeros((E1, F), np.float32)
    return build_tgn(N, E1, D, F, T, k, [0.1], [0.5], w, efeat).eval()


@pytest.mark.parametrize("k,table", [(4, False), (4, True), (20, True), (10, True)])
def test_time_encode_through_embed_kernel(k, table):
    """g6_timeencode (the reference TimeEncode's outputs for dt = 0 ... 3e8, 16777217, ...) through zt_embed:
    the generic kernel without and with the projected table, and the kernel specialised for D = T = 100
    (k = 10, 20: its branch-free cosine and the redo of large-argument columns)."""
    g = golden("g6_timeencode")
    dts = g["dts"].ravel()
    # add the padding-slot value f32(t_now) for t_now as large as SuperUser / the bench stream have
    extra = np.array([6.2e7, 2.4e8, 2.9e8, 4.0e6, 3.99e6, 1.0e7], np.float32)
    tw = g["time_w"]
    all_dt = np.concatenate([dts, extra])
    want = np.concatenate([g["enc"].reshape(-1, 100), np.cos(extra[:, None] * tw[None, :]).astype(np.float32)])
    n = len(all_dt)
    tgn = _time_identity_tgn(50, 10, k)
    dev = tgn.device
    nodes = torch.ones(n, dtype=torch.int32, device=dev)
    on = torch.zeros((1, n, k), dtype=torch.int32, device=dev)
    on[0, :, 2] = 3
    oe = torch.zeros_like(on)
    od = torch.zeros((1, n, k), dtype=torch.float32, device=dev)
    od[0, :, 2] = torch.from_numpy(all_dt).to(dev)
    ow = torch.zeros_like(od)
    ow[0, :, 2] = 1.0
    out = tgn.embedding_module.embed_device(tgn.memory.memory, nodes, on, oe, od, ow,
                                            memory_obj=tgn.memory if table else None).cpu().numpy()
    got = out[:, 100:]
    assert np.abs(got[: len(dts)] - want[: len(dts)]).max() <= 2e-6          # vs the reference's own outputs
    # vs float64 cos of the float32 product (what torch.cos approximates); the ocml cos has its own 1e-7
    x = (all_dt[:, None].astype(np.float32) * tw[None, :]).astype(np.float32).astype(np.float64)
    assert np.abs(got - np.cos(x)).max() <= 2e-6


def test_time_encode_through_message_kernel():
    """The same dt set as message time features: cos((f32(t) - last_update) * w) in zt_store_messages."""
    g = golden("g6_timeencode")
    dts = np.concatenate([g["dts"].ravel(), np.array([6.2e7, 2.4e8, 2.9e8, 4.0e6], np.float32)])
    B = len(dts)
    tgn = _time_identity_tgn(2 * B + 2, B + 2)
    dev = tgn.device
    src = torch.arange(1, B + 1, dtype=torch.int32, device=dev)
    dst = torch.arange(B + 1, 2 * B + 1, dtype=torch.int32, device=dev)
    ts = torch.from_numpy(dts.astype(np.float64)).to(dev)
    eidx = torch.arange(1, B + 1, dtype=torch.int64, device=dev)
    tgn.store_messages_device(src, dst, ts, eidx)
    msg = tgn.memory.messages.cpu().numpy()
    tw = g["time_w"]
    x = (dts[:, None] * tw[None, :]).astype(np.float32).astype(np.float64)
    for rows in (np.arange(1, B + 1), np.arange(B + 1, 2 * B + 1)):
        got = msg[rows, -100:]
        assert np.abs(got - np.cos(x)).max() <= 2e-6
        assert np.abs(got[: g["dts"].size] - g["enc"].reshape(-1, 100)).max() <= 2e-6
    assert np.array_equal(tgn.memory.timestamps.cpu().numpy()[1: B + 1], dts)


def test_protocol_vs_oracle_large_timestamps(oracle):
    """A stream whose clock runs to 3e8 s (SuperUser spans 2.4e8): every padding slot carries dt = f32(t_now)
    and most neighbours dt > 4e6, so the embeddings and messages run on the large-argument cosine."""
    D = T = 100
    N, E, F, bs, k, al, be, seed = 2000, 3600, 1, 300, 20, [0.1, 0.1], [0.5, 0.95], 207
    src, dst, neg, ts, eidx = I.make_stream("bipartite", N, E, seed)
    ts = ts * (3.0e8 / ts[-1])
    w = I.model_weights(D, F, T, 2, seed)
    _, efeat = I.random_tables(N, E + 1, D, F, seed)
    p = oracle.ProtocolOracle(N, D, F, T, k, al, be, w, efeat, I.time_encode_weights(T), n_threads=8)
    tgn = build_tgn(N, E + 1, D, F, T, k, al, be, w, efeat).eval()
    for s in range(0, E, bs):
        e = s + bs
        with torch.no_grad():
            se, de, ne = tgn.compute_temporal_embeddings(src[s:e], dst[s:e], neg[s:e], ts[s:e], eidx[s:e], 10, False)
        emb, _ = p.batch(src[s:e], dst[s:e], neg[s:e], ts[s:e], eidx[s:e], False)
        assert np.abs(torch.cat([se, de, ne]).cpu().numpy() - emb).max() <= TOL, "edge %d" % s
    assert np.abs(tgn.memory.memory.cpu().numpy() - p.mem.memory).max() <= TOL
    assert np.abs(tgn.memory.messages.cpu().numpy() - p.mem.messages).max() <= TOL
    assert np.array_equal(tgn.memory.last_update.cpu().numpy(), p.mem.last_update)


# ---------------------------------------------------------------------------------------------------------
# tppr_strategy='pruning' end to end (config C4's shape): modules/embedding_module.py:221-224,280-297
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", ["eval", "train"])
@pytest.mark.parametrize("name", list(I.PRUNE_EMBED_CASES))
def test_pruning_protocol_golden(name, mode):
    import types
    from zebra_amd.tppr import get_neighbor_finder
    kind, N, E, D, F, T, k, al, be, width, depth, seed, bs, nb, first, n_train = I.PRUNE_EMBED_CASES[name]
    g = golden("g45_prune_" + name)
    src, dst, neg, ts, eidx = I.make_stream(kind, N, E, seed)
    w = I.model_weights(D, F, T, len(al), seed)
    mem0, efeat = I.random_tables(N, E + 1, D, F, seed)
    data = lambda n: types.SimpleNamespace(sources=src[:n], destinations=dst[:n], edge_idxs=eidx[:n], timestamps=ts[:n])
    nf_full, nf_part = get_neighbor_finder(data(E)), get_neighbor_finder(data(n_train))
    tgn = build_tgn(N, E + 1, D, F, T, k, al, be, w, efeat, strategy="pruning", nf=nf_part, width=width, depth=depth)
    train = mode == "train"
    tgn.train(train)
    with torch.no_grad():
        tgn.memory.memory.copy_(torch.from_numpy(mem0))
    for b in range(nb):
        if b == nb // 2:
            tgn.set_neighbor_finder(nf_full)
        s, e = first + b * bs, first + (b + 1) * bs
        ctx = torch.enable_grad() if train else torch.no_grad()
        with ctx:
            se, de, ne = tgn.compute_temporal_embeddings(src[s:e], dst[s:e], neg[s:e], ts[s:e], eidx[s:e], width, train)
            score = tgn.affinity_score(torch.cat([se, se], dim=0), torch.cat([de, ne])).squeeze(dim=0)
        emb = torch.cat([se, de, ne]).detach().cpu().numpy()
        assert np.abs(emb - g["%s_b%d_emb" % (mode, b)]).max() <= TOL, "batch %d" % b
        assert np.abs(score.sigmoid().detach().cpu().numpy().ravel() - g["%s_b%d_prob" % (mode, b)]).max() <= TOL
        if train:
            tgn.memory.detach_memory()
    assert abs(tgn.embedding_module.average_topk - float(g["%s_average_topk" % mode])) < 1e-6
    m = tgn.memory
    pre = "%s_b%d_" % (mode, nb - 1)
    assert np.abs(m.memory.cpu().numpy() - g[pre + "memory"]).max() <= TOL
    assert np.array_equal(m.last_update.cpu().numpy(), g[pre + "last_update"])
    assert np.abs(m.messages.cpu().numpy() - g[pre + "messages"]).max() <= TOL
    assert np.array_equal(m.timestamps.cpu().numpy(), g[pre + "timestamps"])
    assert np.array_equal(m.nodes.astype(np.uint8), g[pre + "flags"])


def test_pruning_protocol_vs_oracle_c4_shape(oracle):
    """C4's shape (SuperUser: non-bipartite, F=1, k=40, width 10, depth 2, bs=1000) on a smaller graph,
    step_device (the path bench.py times) against the oracle protocol."""
    D = T = 100
    N, E, F, bs, k, al, be, seed = 6000, 14000, 1, 1000, 40, [0.1, 0.1], [0.5, 0.95], 208
    import types
    from zebra_amd.tppr import get_neighbor_finder
    src, dst, neg, ts, eidx = I.make_stream("general", N, E, seed)
    w = I.model_weights(D, F, T, 2, seed)
    _, efeat = I.random_tables(N, E + 1, D, F, seed)
    nf = get_neighbor_finder(types.SimpleNamespace(sources=src, destinations=dst, edge_idxs=eidx, timestamps=ts))
    ref_nf = oracle.CsrOracle(src, dst, eidx, ts, N)
    p = oracle.ProtocolOracle(N, D, F, T, k, al, be, w, efeat, I.time_encode_weights(T), "pruning", ref_nf, 10, 2,
                              n_threads=8)
    tgn = build_tgn(N, E + 1, D, F, T, k, al, be, w, efeat, strategy="pruning", nf=nf).eval()
    dev = tgn.device
    t = [torch.from_numpy(x).to(dev) for x in (src, dst, neg, ts, eidx)]
    for s in range(8000, E, bs):
        e = s + bs
        got = tgn.step_device(*[x[s:e] for x in t], check_status=True).cpu().numpy()
        emb, _ = p.batch(src[s:e], dst[s:e], neg[s:e], ts[s:e], eidx[s:e], False)
        assert np.abs(got - emb).max() <= TOL, "edge %d" % s
    assert np.abs(tgn.memory.memory.cpu().numpy() - p.mem.memory).max() <= TOL
    assert np.array_equal(tgn.memory.nodes.astype(np.uint8), p.mem.flags)


# ---------------------------------------------------------------------------------------------------------
# the epoch protocol through the drop-in (train.py:188-191,241-269,296-306)
# ---------------------------------------------------------------------------------------------------------
class _TgnEpochAdapter:
    def __init__(self, tgn):
        self.tgn, self.em = tgn, tgn.embedding_module

    def init_memory(self):
        self.tgn.memory.__init_memory__()

    def reset_tppr(self):
        self.em.reset_tppr()
        self.tgn.set_neighbor_finder(None)

    def fill_tppr(self, src, dst, ts, eidx, filled):
        self.em.fill_tppr(src, dst, ts, eidx, filled)

    def backup_tppr(self):
        return self.em.backup_tppr()

    def restore_tppr(self, b):
        self.em.restore_tppr(b)

    def backup_memory(self):
        return self.tgn.memory.backup_memory()

    def restore_memory(self, b):
        self.tgn.memory.restore_memory(b)

    def batch(self, src, dst, neg, ts, eidx, train):
        self.tgn.train(train)
        with (torch.enable_grad() if train else torch.no_grad()):
            pos, negp = self.tgn.compute_edge_probabilities(src, dst, neg, ts, eidx, 10, train)
        return torch.cat([pos, negp]).detach().cpu().numpy().ravel()

    def memory_state(self):
        m = self.tgn.memory
        return dict(memory=m.memory.detach().cpu().numpy(), last_update=m.last_update.cpu().numpy(),
                    messages=m.messages.detach().cpu().numpy(), timestamps=m.timestamps.cpu().numpy(),
                    flags=m.nodes.astype(np.uint8))

    def tppr_state(self):
        f = self.em.tppr_finder
        return {"m%d_%s" % (m, kk): v for m in range(f.n_tppr) for kk, v in f.export_state(m).items()}


@pytest.mark.parametrize("aliasing", [True, False])
@pytest.mark.parametrize("name", list(I.EPOCH_CASES))
def test_epoch_protocol_golden(name, aliasing):
    """Two epochs of the reference's driver against its own run (g10_epoch).  With
    reference_compat_aliasing the whole protocol must reproduce, including the reference's quirks
    (restore_tppr is a no-op, flags are shared between backups); with deep snapshots (the default)
    everything up to the first restore must."""
    from helpers import epoch_protocol, make_checker
    case = I.EPOCH_CASES[name]
    N, E, D, F, T, k, al, be, seed = case[:9]
    g = golden("g10_epoch_" + name)
    streams = I.make_stream("general", N, E, seed)
    w = I.model_weights(D, F, T, len(al), seed)
    _, efeat = I.random_tables(N, E + 1, D, F, seed)
    tgn = build_tgn(N, E + 1, D, F, T, k, al, be, w, efeat, compat=aliasing)
    assert tgn.memory.reference_compat_aliasing == aliasing
    epoch_protocol(_TgnEpochAdapter(tgn), g, case, streams, make_checker(g, TOL), aliasing=aliasing)


def test_deep_snapshots_restore_what_was_backed_up():
    """Default (deep) snapshots: after backup -> eval batches -> restore, memory, flags and T-PPR state are
    exactly what was backed up (the reference's aliasing loses both, g10 records that)."""
    name = "stream_d20_f7"
    N, E, D, F, T, k, al, be, seed, bs, n_train = I.EPOCH_CASES[name][:11]
    src, dst, neg, ts, eidx = I.make_stream("general", N, E, seed)
    w = I.model_weights(D, F, T, len(al), seed)
    _, efeat = I.random_tables(N, E + 1, D, F, seed)
    tgn = build_tgn(N, E + 1, D, F, T, k, al, be, w, efeat)
    a = _TgnEpochAdapter(tgn)
    for s in range(0, 64, bs):
        a.batch(src[s:s + bs], dst[s:s + bs], neg[s:s + bs], ts[s:s + bs], eidx[s:s + bs], True)
    mem0, tp0 = a.memory_state(), a.tppr_state()
    assert mem0["flags"].sum() > 0
    mb, tb = a.backup_memory(), a.backup_tppr()
    for s in range(64, 128, bs):
        a.batch(src[s:s + bs], dst[s:s + bs], neg[s:s + bs], ts[s:s + bs], eidx[s:s + bs], False)
    assert not np.array_equal(a.memory_state()["memory"], mem0["memory"])
    a.restore_memory(mb)
    a.restore_tppr(tb)
    for kk, v in a.memory_state().items():
        assert np.array_equal(v, mem0[kk]), kk
    for kk, v in a.tppr_state().items():
        assert np.array_equal(v, tp0[kk]), kk


# ---------------------------------------------------------------------------------------------------------
# projected memory table (zt_project_memory): same embeddings as the full contraction, and it follows every
# way the memory or the weights can change
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("F", [1, 172])
def test_projected_table_follows_memory_and_weights(oracle, F):
    D = T = 100
    N, E, bs, k, al, be, seed = 1500, 3000, 250, 20, [0.1, 0.1], [0.5, 0.95], 209
    src, dst, neg, ts, eidx = I.make_stream("bipartite", N, E, seed)
    w = I.model_weights(D, F, T, 2, seed)
    mem0, efeat = I.random_tables(N, E + 1, D, F, seed)
    tw = I.time_encode_weights(T)
    a = build_tgn(N, E + 1, D, F, T, k, al, be, w, efeat).eval()        # projected table
    b = build_tgn(N, E + 1, D, F, T, k, al, be, w, efeat).eval()        # full contraction
    b.embedding_module.use_projection = False
    p = oracle.ProtocolOracle(N, D, F, T, k, al, be, w, efeat, tw, n_threads=8)
    dev = a.device

    def step(s, e):
        outs = []
        for t in (a, b):
            with torch.no_grad():
                se, de, ne = t.compute_temporal_embeddings(src[s:e], dst[s:e], neg[s:e], ts[s:e], eidx[s:e], 10, False)
            outs.append(torch.cat([se, de, ne]).cpu().numpy())
        want, _ = p.batch(src[s:e], dst[s:e], neg[s:e], ts[s:e], eidx[s:e], False)
        assert np.abs(outs[0] - want).max() <= TOL and np.abs(outs[1] - want).max() <= TOL, "edge %d" % s
        assert np.abs(outs[0] - outs[1]).max() <= 2e-5

    for s in range(0, 1000, bs):
        step(s, s + bs)
    assert a.embedding_module._proj is not None and b.embedding_module._proj is None
    # 1. torch-side in-place change of the memory (version bump) -> rebuilt
    with torch.no_grad():
        for t in (a, b):
            t.memory.memory.copy_(torch.from_numpy(mem0).to(dev))
    p.mem.memory[...] = mem0
    step(1000, 1250)
    # 2. set_memory on a few rows
    ids = np.array([3, 17, 400], np.int64)
    vals = torch.full((3, D), 0.25, device=dev)
    for t in (a, b):
        t.memory.set_memory(ids, vals)
    p.mem.memory[ids] = 0.25
    step(1250, 1500)
    # 3. backup / restore (new tensors)
    bk = [t.memory.backup_memory() for t in (a, b)]
    pb = p.backup_memory()
    step(1500, 1750)
    for t, x in zip((a, b), bk):
        t.memory.restore_memory(x)
    p.restore_memory(pb)
    p.mem.flags = p.mem.flags.copy()
    p.mem.flags[...] = 0                                   # eval mode leaves no pending messages; deep copies agree
    step(1500, 1750)
    # 4. the weights change in place (what an optimizer step does)
    with torch.no_grad():
        for t in (a, b):
            t.embedding_module.fc1.weight.mul_(0.5)
            t.embedding_module.fc2_source.bias.add_(0.125)
    p.w = dict(p.w)
    p.w["fc1_w"] = p.w["fc1_w"] * np.float32(0.5)
    p.w["fc2s_b"] = p.w["fc2s_b"] + np.float32(0.125)
    step(1750, 2000)
    # 5. __init_memory__ (new epoch)
    for t in (a, b):
        t.memory.__init_memory__()
    p.init_memory()
    step(2000, 2250)


@pytest.mark.parametrize("F,k", [(1, 20), (172, 20), (4, 40), (1, 64)])       # (k = 64: the wide streaming path under training)
def test_fused_training_backward_full_dims(F, k):
    """The fused HIP training path (overlay forward + k_fc1_agg_bwd) at the real layer sizes (D = T = 100,
    two T-PPR models) against the torch composition of the same step, which fixture g8_train_grads pins to the
    reference at small sizes.  The cotangent of the embeddings is a fixed random matrix (a linear loss), so
    that nothing downstream -- a ReLU of the scorer sitting at zero -- can amplify rounding differences:
    embeddings and the gradients of all embedding / GRU parameters must agree over dependent batches."""
    D = T = 100
    N, E, bs, al, be, seed = 900, 2400, 200, [0.1, 0.1], [0.5, 0.95], 210
    src, dst, neg, ts, eidx = I.make_stream("bipartite", N, E, seed)
    w = I.model_weights(D, F, T, 2, seed)
    _, efeat = I.random_tables(N, E + 1, D, F, seed)
    dev = torch.device("cuda")
    G = [torch.from_numpy(np.random.RandomState(700 + b).standard_normal((3 * bs, 3 * D)).astype(np.float32)).to(dev)
         for b in range(6)]
    res = {}
    for fused in (True, False):
        tgn = build_tgn(N, E + 1, D, F, T, k, al, be, w, efeat)
        tgn.embedding_module.fused_training = fused
        tgn.train(True)
        out = []
        for b in range(6):
            s, e = b * bs, (b + 1) * bs
            tgn.zero_grad()
            se, de, ne = tgn.compute_temporal_embeddings(src[s:e], dst[s:e], neg[s:e], ts[s:e], eidx[s:e], 10, True)
            emb = torch.cat([se, de, ne])
            (emb * G[b]).sum().backward()
            out.append((emb.detach().cpu().numpy(), {pn: p.grad.detach().cpu().numpy().copy()
                                                    for pn, p in tgn.named_parameters() if p.grad is not None}))
        res[fused] = out
    for b in range(6):
        ea, ga = res[True][b]
        eb, gb = res[False][b]
        assert np.abs(ea - eb).max() <= 1e-5, "embeddings of batch %d" % b
        assert set(ga) == set(gb) and len(ga) >= 8
        for pn in ga:
            # float32 sums in another order, and now and then a ReLU whose pre-activation is within rounding of
            # zero (the two paths round the fc1 product differently): a flipped unit changes one row of a gradient
            # by one term.  So: all but a sliver of the elements to 1e-4 of the scale, the whole to 2e-3 in norm.
            d, scale = np.abs(ga[pn] - gb[pn]), max(1.0, np.abs(gb[pn]).max())
            assert (d > 1e-4 * scale).mean() <= 0.01, "%s in batch %d: %.3g of the elements differ" % (pn, b, (d > 1e-4 * scale).mean())
            assert np.linalg.norm(ga[pn] - gb[pn]) <= 2e-3 * max(1.0, np.linalg.norm(gb[pn])), "%s in batch %d" % (pn, b)
    assert any(np.abs(res[True][b][1]["memory_updater.memory_updater.weight_ih"]).max() > 0 for b in range(1, 6))


def test_overlay_rows_forward_and_backward():
    """zt_overlay_rows: the batch's own rows of the lazily updated memory (get_updated_memory(...)[nodes],
    modules/memory_updater.py:61-90) and the gradient to the overlay rows, against torch's index / where composition --
    nodes that repeat within the batch, nodes without an overlay row, and no overlay at all."""
    from zebra_amd.modules import _OverlayRows
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(5)
    N, D, U, n = 500, 100, 37, 600
    memory = torch.randn((N, D), generator=g).to(dev)
    ids = torch.randperm(N, generator=g)[:U].to(dev)
    row_map = torch.full((N,), -1, dtype=torch.int32, device=dev)
    row_map[ids] = torch.arange(U, dtype=torch.int32, device=dev)
    nodes = torch.randint(0, N, (n,), generator=g).to(dev)
    nodes[:50] = ids[torch.randint(0, U, (50,), generator=g).to(dev)]          # repeated overlay rows
    cot = torch.randn((n, D), generator=g).to(dev)
    ov_a = torch.randn((U, D), generator=g).to(dev).requires_grad_(True)
    ov_b = ov_a.detach().clone().requires_grad_(True)
    out = _OverlayRows.apply(ov_a, memory, row_map, nodes.to(torch.int32), True)
    (out * cot).sum().backward()
    m = row_map[nodes].long()
    ref = torch.where((m >= 0).unsqueeze(1), ov_b[m.clamp(min=0)], memory[nodes])
    (ref * cot).sum().backward()
    assert torch.equal(out, ref)
    assert (m >= 0).sum() >= 50 and (m < 0).sum() > 0
    assert torch.allclose(ov_a.grad, ov_b.grad, rtol=0, atol=1e-5)           # (sums of a few float32 terms in another order)
    none = _OverlayRows.apply(torch.zeros((1, D), device=dev), memory, row_map, nodes.to(torch.int32), False)
    assert torch.equal(none, memory[nodes])


@pytest.mark.parametrize("F,k", [(1, 20), (4, 10)])
def test_fused_training_dropout(F, k):
    """The reference trains with nn.Dropout(0.1) between fc1's ReLU and fc2 (modules/embedding_module.py:89,
    323-326).  The fused kernels draw the keep-mask from a hash of (seed, element): forward values and all
    gradients must equal the torch composition multiplied by that very mask (zebra_amd.modules.dropout_mask is
    the hash in numpy), the mask must keep ~90 % and differ from seed to seed, and p = 0 must be the old path."""
    from zebra_amd.modules import _NeighbourAggregate, dropout_mask
    D = T = 100
    N, E1, n, M, p = 700, 3000, 257, 2, 0.1
    g = torch.Generator().manual_seed(41 + F)
    w = I.model_weights(D, F, T, M, 55)
    _, efeat = I.random_tables(N, E1, D, F, 55)
    tgn = build_tgn(N, E1, D, F, T, k, [0.1, 0.1], [0.5, 0.95], w, efeat)
    em = tgn.embedding_module
    dev = tgn.device
    mem = torch.randn((N, D), generator=g).to(dev)
    U = 40
    ids = torch.randperm(N, generator=g)[:U].to(dev)
    overlay = torch.randn((U, D), generator=g).to(dev).requires_grad_(True)
    row_map = torch.full((N,), -1, dtype=torch.int32, device=dev)
    row_map[ids] = torch.arange(U, dtype=torch.int32, device=dev)
    on = torch.randint(0, N, (M, n, k), generator=g, dtype=torch.int32).to(dev)
    oe = torch.randint(0, E1, (M, n, k), generator=g, dtype=torch.int32).to(dev)
    od = (torch.rand((M, n, k), generator=g) * 1e5).to(dev)
    ow = torch.rand((M, n, k), generator=g)
    ow[:, ::6] = 0.0
    ow = ow.to(dev)
    G = torch.randn((M, n, D), generator=g).to(dev)
    fc1_w = em.fc1.weight.detach().clone().requires_grad_(True)
    fc1_b = em.fc1.bias.detach().clone().requires_grad_(True)

    def fused(seed, pp):
        for t in (overlay, fc1_w, fc1_b):
            t.grad = None
        em.fc1.weight.data.copy_(fc1_w.data); em.fc1.bias.data.copy_(fc1_b.data)
        H, S = _NeighbourAggregate.apply(overlay, fc1_w, fc1_b, em, mem, row_map, ids.to(torch.int32), on, oe, od, ow, pp, seed)
        (H * G).sum().backward()
        row_map[ids] = torch.arange(U, dtype=torch.int32, device=dev)       # (the backward resets the shared map)
        return H.detach().cpu().numpy(), [t.grad.detach().cpu().numpy().copy() for t in (overlay, fc1_w, fc1_b)]

    def composed(mask):
        for t in (overlay, fc1_w, fc1_b):
            t.grad = None
        rows = torch.where((row_map[on.long()] >= 0).unsqueeze(-1), overlay[row_map[on.long()].long().clamp(min=0)], mem[on.long()])
        x = torch.cat([rows, em.edge_features[oe.long()], em.time_encoder(od.reshape(M * n, k)).reshape(M, n, k, T)], dim=-1)
        h = torch.relu(torch.nn.functional.linear(x, fc1_w, fc1_b)) * mask
        ws = ow.sum(dim=2, keepdim=True)
        wn = torch.where(ws == 0, torch.zeros_like(ow), ow / ws)
        H = (h * wn.unsqueeze(-1)).sum(dim=2)
        (H * G).sum().backward()
        return H.detach().cpu().numpy(), [t.grad.detach().cpu().numpy().copy() for t in (overlay, fc1_w, fc1_b)]

    seed = 0x1234567811223344
    mask = dropout_mask(seed, p, (M, n, k), D)
    assert abs((mask > 0).mean() - (1 - p)) < 0.005 and np.allclose(mask[mask > 0], 1 / (1 - p))
    assert (dropout_mask(seed + 1, p, (M, n, k), D) != mask).mean() > 0.1
    Hf, gf = fused(seed, p)
    Hc, gc = composed(torch.from_numpy(mask).to(dev))
    assert np.abs(Hf - Hc).max() <= 2e-5 * max(1.0, np.abs(Hc).max())
    for a, b in zip(gf, gc):
        assert np.linalg.norm(a - b) <= 2e-4 * max(1.0, np.linalg.norm(b))
    H0, g0 = fused(0, 0.0)
    H1, g1 = composed(torch.ones((M, n, k, D), device=dev))
    assert np.abs(H0 - H1).max() <= 2e-5 * max(1.0, np.abs(H1).max())
    assert np.abs(Hf - H0).max() > 1e-3                                       # the mask does something


@pytest.mark.parametrize("F,k,n", [(1, 20, 1001), (172, 10, 333), (4, 40, 257), (1, 40, 2),
                                   (172, 20, 1001), (172, 20, 3), (172, 40, 257), (172, 40, 1), (172, 20, 7201)])
def test_specialised_aggregate_equals_generic(F, k, n):
    """The specialised aggregate kernels -- k_fc1_agg_reg (F <= 4, k in {20, 40}), k_fc1_agg_wide (F = 172, k in
    {20, 40}: weights resident in LDS, M-tile units, group partial sums), k_fc1_agg_d100 (the other D = T = 100
    shapes) -- against the generic table kernel (zt_set_kernel_choice) on the same inputs: ragged last tile, empty rows
    (all-zero weights), time gaps on both sides of the 4e6 switch."""
    import os
    D = T = 100
    N, E1 = 5000, 20000
    g = torch.Generator().manual_seed(11 + F + k)
    w = I.model_weights(D, F, T, 2, 77)
    _, efeat = I.random_tables(N, E1, D, F, 77)
    tgn = build_tgn(N, E1, D, F, T, k, [0.1, 0.1], [0.5, 0.95], w, efeat).eval()
    dev = tgn.device
    tgn.memory.memory.copy_(torch.randn((N, D), generator=g).to(dev))
    nodes = torch.randint(0, N, (n,), generator=g, dtype=torch.int32).to(dev)
    on = torch.randint(0, N, (2, n, k), generator=g, dtype=torch.int32)
    oe = torch.randint(0, E1, (2, n, k), generator=g, dtype=torch.int32)
    od = torch.rand((2, n, k), generator=g) * 3.0e6
    od[:, ::7] *= 1.0e3                                   # some rows far beyond the fast cosine's range
    ow = torch.rand((2, n, k), generator=g)
    ow[:, ::5] = 0.0                                      # rows whose weights sum to 0
    ow[:, 1::5, k // 2:] = 0.0                            # short dictionaries
    args = [t.to(dev).contiguous() for t in (on, oe, od.float(), ow.float())]
    em = tgn.embedding_module
    fast = em.embed_device(tgn.memory.memory, nodes, *args, memory_obj=tgn.memory).cpu().numpy()
    from zebra_amd import _capi
    _capi.set_kernel_choice(_capi.CHOICE_AGGREGATE, _capi.AGG_GENERIC)
    try:
        slow = em.embed_device(tgn.memory.memory, nodes, *args, memory_obj=tgn.memory).cpu().numpy()
    finally:
        _capi.set_kernel_choice(_capi.CHOICE_AGGREGATE, 0)
    plain = em.embed_device(tgn.memory.memory, nodes, *args).cpu().numpy()          # no table at all
    assert np.abs(fast - slow).max() <= 2e-5 * max(1.0, np.abs(slow).max())
    assert np.abs(fast - plain).max() <= 1e-4 * max(1.0, np.abs(plain).max())


def test_randomised_pipeline_soak(oracle):
    """A slice of tests/soak_pipeline.py: random node counts, batch sizes (ragged last batches), k, layer widths (every
    aggregation kernel), feature widths, launch groups, CU masks, views ahead, shuffled ids -- the whole step through the
    native pipeline against the oracle's protocol (embeddings <= 1e-4 per batch, T-PPR state bit-exact, memory tables)."""
    import soak_pipeline
    for seed in range(9000, 9030):
        err = soak_pipeline.one(seed, torch, oracle)
        assert err is None, err


def test_randomised_training_soak():
    """A slice of tests/soak_train.py: the fused HIP training path (aggregation over the overlay, GRU rows, fc2, source
    transform; forward and backward) against the torch composition on random shapes -- embeddings to 1e-5, every parameter
    gradient to 1e-4 of its scale (ReLU units flipped by rounding are identified and set aside)."""
    import soak_train
    for seed in range(30000, 30060):
        err = soak_train.one(seed, torch)
        assert err is None, err
