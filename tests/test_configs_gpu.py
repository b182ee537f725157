"""Every BASELINE.json config at its REAL shape against the CPU oracle, through the path bench.py times.

``zebra_amd.synth.WORKLOADS`` c1..c5 (SURVEY.md section 8, table at the top; reference
train.py:22-59,124-139) at FULL node / edge counts -- C1/C2 9 228 nodes F = 172 bs = 200, C3 10 985 /
F = 172 / bs = 600, C4 194 086 / k = 40 / pruning (10, 2) / bs = 1000, C5 10 000 001 / bs = 4096 -- stepped with
``TGN.step_device`` through ``enable_pipeline`` with exactly the (tppr_cus, group) that bench.py chooses
(``synth.pipeline_settings``: the one function both call), node ids shuffled as in the bench, the batches in sight
the bench's ``look``.  A prefix is run unchecked (bench.py's own prefill length where the oracle affords it), then
>= 30 batches are compared: embeddings of every batch <= 1e-4, at the end the T-PPR state bit-exact and the memory
tables (memory, last_update, messages, message time stamps, flags).  (reference utils/util.py:473-576 for the
streaming update, :185-276 for the pruning one, model/tgn_model.py:124-174 for the step.)
"""
import types

import numpy as np
import pytest
import torch

import inputs as I
from helpers import build_tgn

pytestmark = pytest.mark.gpu
TOL = 1e-4
D = T = 100


def _stream(wl, n_edges, perm_seed=7, seed=2020):
    """bench.py's make_stream: the same generator, seeds and id shuffle."""
    from zebra_amd import synth
    src, dst, ts, eidx = synth.power_law_stream(wl["n_nodes"], n_edges, bipartite=wl["bipartite"], seed=seed,
                                                perm_seed=perm_seed)
    neg = synth.negatives(dst, n_edges, seed=seed + 1)
    return src, dst, neg, ts, eidx


def _compare_tables(tgn, p, ids=None):
    """memory / last_update / messages / message time stamps / flags; ``ids``: these rows only (full-size C5)."""
    m = tgn.memory
    if ids is None:
        sel = lambda t: t.cpu().numpy()
        osel = lambda a: a
    else:
        ids_d = torch.from_numpy(ids).to(tgn.device)
        sel = lambda t: t.index_select(0, ids_d).cpu().numpy()
        osel = lambda a: a[ids]
    assert np.abs(sel(m.memory) - osel(p.mem.memory)).max() <= TOL
    assert np.array_equal(sel(m.last_update), osel(p.mem.last_update))
    assert np.abs(sel(m.messages) - osel(p.mem.messages)).max() <= TOL
    assert np.array_equal(sel(m.timestamps), osel(p.mem.timestamps))
    assert np.array_equal(sel(m.flags), osel(p.mem.flags))


def _run(oracle, name, first_batch, fill, nb, steps_for_group, group=-1, tppr_cus=-1, csr_edges=None, touched_only=False,
         wl=None, perm=7, by_launch=False):
    """Step batches first_batch .. first_batch+fill+nb of workload ``name`` through the pipeline and the oracle.
    by_launch: the aggregation of a batch waits for the END of the T-PPR launch its group shares (ZT_RELEASE_LAUNCH: the form
    before round 6, whose groups depend on the length of the timed region) instead of for that batch's rows."""
    from zebra_amd import _capi
    _capi.set_kernel_choice(_capi.CHOICE_GROUP_RELEASE, _capi.RELEASE_LAUNCH if by_launch else 0)
    try:
        return _run_inner(oracle, name, first_batch, fill, nb, steps_for_group, group, tppr_cus, csr_edges, touched_only, wl, perm)
    finally:
        _capi.set_kernel_choice(_capi.CHOICE_GROUP_RELEASE, 0)


def _run_inner(oracle, name, first_batch, fill, nb, steps_for_group, group, tppr_cus, csr_edges, touched_only, wl, perm):
    from zebra_amd import synth
    from zebra_amd.tppr import get_neighbor_finder
    wl = wl or synth.WORKLOADS[name]
    bs, k, F, al, be = wl["bs"], wl["k"], wl["F"], wl["alpha"], wl["beta"]
    M = len(al)
    N = wl["n_nodes"] + 1
    n_b = first_batch + fill + nb
    E = csr_edges if csr_edges is not None else n_b * bs
    assert E <= wl["n_edges"] and n_b * bs <= E
    src, dst, neg, ts, eidx = _stream(wl, E, perm_seed=perm)
    w = I.model_weights(D, F, T, M, 404)
    E1 = (wl["n_edges"] if F == 1 else E) + 1                  # F = 1: the full |E|+1 zero table, as in bench.py
    efeat = synth.edge_features(E1, F, seed=405)
    tw = I.time_encode_weights(T)
    nf = ref_nf = None
    if wl["strategy"] == "pruning":
        nf = get_neighbor_finder(types.SimpleNamespace(sources=src, destinations=dst, edge_idxs=eidx, timestamps=ts))
        ref_nf = oracle.CsrOracle(src, dst, eidx, ts, N)
    tgn = build_tgn(N, E1, D, F, T, k, al, be, w, efeat, strategy=wl["strategy"], nf=nf, width=wl.get("width", 10),
                    depth=wl.get("depth", 2)).eval()
    cus, grp = synth.pipeline_settings(wl, steps_for_group, tppr_cus, group)
    tgn.enable_pipeline(tppr_cus=cus, group=grp)
    p = oracle.ProtocolOracle(N, D, F, T, k, al, be, w, efeat, tw, wl["strategy"], ref_nf, wl.get("width", 10),
                              wl.get("depth", 2), n_threads=8)
    dev = tgn.device
    t = [torch.from_numpy(x).to(dev) for x in (src, dst, neg, ts, eidx)]
    batches = [tuple(x[b * bs:(b + 1) * bs] for x in t) for b in range(first_batch, n_b)]
    look = synth.pipeline_look(grp)                            # bench.py's view ahead
    embs = []
    with torch.cuda.stream(tgn.main_stream):
        for q, cur in enumerate(batches):
            e = tgn.step_device(*cur, ahead=batches[q + 1: q + 1 + look])
            if q >= fill:
                embs.append(e.clone())
    torch.cuda.synchronize()
    if wl["strategy"] == "streaming":
        tgn.embedding_module.tppr_finder.check_status()
    st = int(tgn.embedding_module._status.item()) if tgn.embedding_module._status is not None else 0
    assert st == 0, "device status %d" % st
    worst = 0.0
    for q in range(fill + nb):
        s, e = (first_batch + q) * bs, (first_batch + q + 1) * bs
        ref, _ = p.batch(src[s:e], dst[s:e], neg[s:e], ts[s:e], eidx[s:e], False)
        if q >= fill:
            d = float(np.abs(embs[q - fill].cpu().numpy() - ref).max())
            assert d <= TOL, "%s: embeddings of batch %d differ from the oracle by %g" % (name, first_batch + q, d)
            worst = max(worst, d)
    s0, s1 = first_batch * bs, n_b * bs
    ids = np.unique(np.concatenate([src[s0:s1], dst[s0:s1], neg[s0:s1]])).astype(np.int64) if touched_only else None
    if wl["strategy"] == "streaming":
        f = tgn.embedding_module.tppr_finder
        for m in range(M):
            a, bb = (f.export_rows(m, ids), p.tppr.export_rows(m, ids)) if touched_only else (f.export_state(m), p.tppr.export(m))
            for kk in a:
                assert np.array_equal(a[kk], bb[kk]), "%s: T-PPR state %s of model %d differs from the oracle" % (name, kk, m)
    _compare_tables(tgn, p, ids)
    tgn.enable_pipeline(False)
    return worst, cus, grp


@pytest.mark.parametrize("name,steps", [("c1", 200), ("c2", 200), ("c2", 20)])
def test_wikipedia_configs_vs_oracle(oracle, name, steps):
    """C1 (one T-PPR model, alpha 0.1 / beta 0.9: train.py:57) and C2 (two models) at Wikipedia's shape: 9 228 node
    ids, bipartite 8 227 + 1 000, F = 172, bs = 200, k = 20; bench.py's prefill (10 % of the stream = 78 batches), then
    32 checked batches; launch groups of 8 batches released to the aggregation one by one (what bench.py runs whatever the
    length of the region, round 6) and, steps = 20, the groups of 2 that the release by launch picks for the driver's 20-step
    run."""
    from zebra_amd import synth
    wl = synth.WORKLOADS[name]
    fill = (wl["n_edges"] // 10) // wl["bs"]
    worst, cus, grp = _run(oracle, name, 0, fill, 32, steps, by_launch=steps < 80)
    assert cus == 64 and grp == (8 if steps >= 80 else 2)


@pytest.mark.parametrize("steps", [200, 20])
def test_reddit_config_vs_oracle(oracle, steps):
    """C3: Reddit's shape -- 10 985 node ids, bipartite 10 000 + 984, F = 172, bs = 600 (1 800 rows), k = 20, two
    models -- the whole step: bench.py's prefill (112 batches), then 30 checked batches.  The combination the round-3
    review found in no test."""
    from zebra_amd import synth
    wl = synth.WORKLOADS["c3"]
    fill = (wl["n_edges"] // 10) // wl["bs"]
    worst, cus, grp = _run(oracle, "c3", 0, fill, 30, steps, by_launch=steps < 80)
    assert cus == 64 and grp == (8 if steps >= 80 else 2)


def test_superuser_config_vs_oracle(oracle):
    """C4: SuperUser's shape -- 194 086 node ids, the static adjacency of ALL 1 443 339 edges (2.9 M CSR entries),
    F = 1, k = 40, pruning (width 10, depth 2), bs = 1000 -- stepped where the test split would be (from edge
    1 000 000: neighbourhoods are full there), 8 unchecked + 30 checked batches."""
    from zebra_amd import synth
    wl = synth.WORKLOADS["c4"]
    worst, cus, grp = _run(oracle, "c4", 1000, 8, 30, 200, csr_edges=wl["n_edges"])
    assert cus == 0 and grp == 1


@pytest.mark.parametrize("group,perm", [(2, 7), (4, 7), (2, None), (4, None)])
def test_bench_path_vs_oracle_c5_launch_shapes(oracle, group, perm):
    """C5's launch configuration as bench.py runs it -- T-PPR on 64 CUs (two XCDs; the prepass sizes the number of hub
    chains by the general queue's load), launches over 2 (the driver's 20-step run) and 4 batches (200 steps) -- on 100 K
    nodes, where the hub's chain is a tenth of the batch: 4 unchecked + 12 checked batches; ids shuffled as in the bench and
    id == popularity rank (the hot rows contiguous)."""
    from zebra_amd import synth
    wl = dict(synth.WORKLOADS["c5"], n_nodes=100_000, n_edges=16 * 4096)
    steps = 20 if group == 2 else 200
    worst, cus, grp = _run(oracle, "c5 on 100 K nodes", 0, 4, 12, steps, wl=wl, perm=perm, by_launch=group == 2)
    assert (cus, grp) == (64, group)


def test_wiki_talk_scale_config_vs_oracle(oracle):
    """C5 at FULL size: 10 000 001 node ids (T-PPR state 19.8 GB, memory tables 16 GB), bs = 4096, k = 20, two models,
    the driver's launch shape (64 CUs, four batches per launch, released to the aggregation one by one): 6 unchecked + 30
    checked batches against the oracle; state and tables compared on the rows the stream touched (the others were never written
    on either side)."""
    worst, cus, grp = _run(oracle, "c5", 0, 6, 30, 20, touched_only=True)
    assert (cus, grp) == (64, 4)
