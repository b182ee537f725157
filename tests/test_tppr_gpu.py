"""GPU parity of the HIP T-PPR kernels: bit-exact against the golden vectors
(produced by the reference) and against the CPU oracle on larger seeded
streams.  All calls go through the C-ABI (zebra_amd._capi -> libzebra_amd.so)."""
import types

import numpy as np
import pytest

import inputs as I
from conftest import golden

pytestmark = pytest.mark.gpu


def _variant_choice(which, value):
    """The chain modes / prepass forms that were measured slower than the library's pick are compiled into VARIANT builds only
    (tools/build_variant.sh, sources under tools/exp/variants/): the product library refuses them, and their tests run with
    ZT_TEST_LIB=tools/out/<variant>/libzebra_amd.so (tests/conftest.py)."""
    from zebra_amd import _capi
    try:
        _capi.set_kernel_choice(which, value)
    except ValueError as exc:
        pytest.skip("variant build only: %s" % exc)


@pytest.fixture(scope="module")
def zt():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from zebra_amd import tppr
    return tppr


def _cmp_state(a, b, what=""):
    for kk in ("len", "norm", "eidx", "node", "ts", "w"):
        assert np.array_equal(a[kk], b[kk]), "state %s differs %s" % (kk, what)


@pytest.mark.parametrize("name", list(I.STREAM_CASES))
def test_streaming_golden(zt, name):
    kind, N, E, seed, bs, k, al, be, _ = I.STREAM_CASES[name]
    g = golden("g1_stream_" + name)
    src, dst, neg, ts, eidx = I.make_stream(kind, N, E, seed)
    f = zt.tppr_finder(N, k, len(al), al, be)
    nb = (E + bs - 1) // bs
    full = set(g["full_idx"].tolist())
    for b in range(nb):
        s, e = b * bs, min(E, (b + 1) * bs)
        nodes = np.concatenate([src[s:e], dst[s:e], neg[s:e]])
        on, oe, od, ow = f.streaming_topk(nodes, np.concatenate([ts[s:e]] * 3), eidx[s:e])
        if b in full:
            assert np.array_equal(np.stack(on), g["b%d_nodes" % b])
            assert np.array_equal(np.stack(oe), g["b%d_eidx" % b])
            assert np.array_equal(np.stack(od), g["b%d_dt" % b])
            assert np.array_equal(np.stack(ow), g["b%d_w" % b])
        arrs = []
        for m in range(len(al)):
            arrs += [on[m], oe[m], od[m], ow[m]]
        assert I.digest(arrs) == str(g["digests"][b]), "batch %d differs from the reference" % b
    for m in range(len(al)):
        st = f.export_state(m)
        gs = {kk: g["state%d_%s" % (m, kk)] for kk in st}
        _cmp_state(st, gs, "model %d" % m)


@pytest.mark.parametrize("name", [n for n in I.STREAM_CASES if I.STREAM_CASES[n][5] <= 20])
def test_streaming_golden_paired_hops(zt, name):
    """The same goldens with hub chains taking TWO positions per critical section where the preconditions hold
    (csrc/tppr_pair.hpp; zt_set_kernel_choice(ZT_CHOICE_TPPR_CHAIN, ZT_CHAIN_PAIRED); k <= 20): outputs and state equal the
    reference's bit for bit -- the hub-tie stream included --, and on the streams with a hub pairs do form."""
    from zebra_amd import _capi
    _variant_choice(_capi.CHOICE_TPPR_CHAIN, _capi.CHAIN_PAIRED)
    try:
        test_streaming_golden(zt, name)
    finally:
        _capi.set_kernel_choice(_capi.CHOICE_TPPR_CHAIN, 0)


@pytest.mark.parametrize("name", list(I.STREAM_CASES))
def test_streaming_golden_duo(zt, name):
    """... and in duo mode (ZT_CHAIN_DUO: the weights' recurrence on a wave of its own, running ahead of the spine on the
    assumption that every test of the lean section passes; the spine voids its records whenever one does not)."""
    from zebra_amd import _capi
    _variant_choice(_capi.CHOICE_TPPR_CHAIN, _capi.CHAIN_DUO)
    try:
        test_streaming_golden(zt, name)
    finally:
        _capi.set_kernel_choice(_capi.CHOICE_TPPR_CHAIN, 0)


@pytest.mark.parametrize("name", list(I.STREAM_CASES))
def test_streaming_golden_spine(zt, name):
    """The same goldens with hub chains in spine mode (csrc/tppr_chain.hpp: one wave per chain runs every critical section
    with the hub's row in registers, the others prepare and finish; ZT_CHAIN_SPINE): bit for bit the reference's."""
    from zebra_amd import _capi
    _variant_choice(_capi.CHOICE_TPPR_CHAIN, _capi.CHAIN_SPINE)
    try:
        test_streaming_golden(zt, name)
    finally:
        _capi.set_kernel_choice(_capi.CHOICE_TPPR_CHAIN, 0)


def test_duo_dense_hub_and_soak(zt, oracle):
    """Duo mode on the dense-hub graphs and a slice of the randomised soak (restarts of the weights wave: every position the
    spine leaves to a helper voids its records)."""
    from zebra_amd import _capi
    import soak_tppr
    _variant_choice(_capi.CHOICE_TPPR_CHAIN, _capi.CHAIN_DUO)
    try:
        for seed in (1, 2):
            test_dense_hub_graph_vs_oracle(zt, oracle, seed)
        for seed in range(54000, 54040):
            err = soak_tppr.one(seed, zt, oracle)
            assert err is None, err
    finally:
        _capi.set_kernel_choice(_capi.CHOICE_TPPR_CHAIN, 0)


def test_spine_dense_hub_and_soak(zt, oracle):
    """Spine mode on the dense-hub graphs (long chains, hub-hub edges, exact ties), a slice of the randomised soak and a hub
    stream long enough for chains, against the oracle; the chain statistics show the spine running the sections."""
    from zebra_amd import _capi
    import soak_tppr
    _variant_choice(_capi.CHOICE_TPPR_CHAIN, _capi.CHAIN_SPINE)
    try:
        for seed in (1, 2):
            test_dense_hub_graph_vs_oracle(zt, oracle, seed)
        for seed in range(53000, 53040):
            err = soak_tppr.one(seed, zt, oracle)
            assert err is None, err
        N, E, k, bs = 400, 8000, 20, 2000
        rng = np.random.RandomState(5)
        src = np.where(rng.rand(E) < 0.2, 7, rng.randint(1, N, E)).astype(np.int32)
        dst = rng.randint(1, N, E).astype(np.int32)
        neg = rng.randint(1, N, E).astype(np.int32)
        ts = np.cumsum(rng.rand(E) * 10.0)
        eidx = np.arange(1, E + 1, dtype=np.int64)
        f = zt.tppr_finder(N, k, 2, [0.1, 0.1], [0.5, 0.95])
        o = oracle.TpprOracle(N, k, 2, [0.1, 0.1], [0.5, 0.95])
        f.chain_stats()
        for s in range(0, E, bs):
            nodes = np.concatenate([src[s:s + bs], dst[s:s + bs], neg[s:s + bs]])
            a = f.streaming_topk(nodes, ts[s:s + bs], eidx[s:s + bs])
            b = o.streaming_topk(nodes, ts[s:s + bs], eidx[s:s + bs])
            for x, y in zip(a, b):
                assert np.array_equal(np.stack(x), np.stack(y))
        st = f.chain_stats()
        assert st["pairs_done"] > st["pairs_left_in_section"] > 0, st      # (spine mode: sections run by the spine / left to helpers)
        for m in range(2):
            _cmp_state(f.export_state(m), o.export(m), "model %d" % m)
    finally:
        _capi.set_kernel_choice(_capi.CHOICE_TPPR_CHAIN, 0)


def test_paired_hops_dense_hub_and_soak(zt, oracle):
    """Paired chain hops on the dense-hub graphs (every edge touches one of a few hubs: long chains, hub-hub edges, exact
    ties) and a slice of the randomised soak, against the oracle; the chain statistics show pairs completing."""
    from zebra_amd import _capi
    import soak_tppr
    _variant_choice(_capi.CHOICE_TPPR_CHAIN, _capi.CHAIN_PAIRED)
    try:
        for seed in (1, 2):
            test_dense_hub_graph_vs_oracle(zt, oracle, seed)
        for seed in range(52000, 52040):
            err = soak_tppr.one(seed, zt, oracle)
            assert err is None, err
        # a hub stream long enough for chains: pairs must actually form
        N, E, k, bs = 400, 8000, 20, 2000
        rng = np.random.RandomState(5)
        src = np.where(rng.rand(E) < 0.2, 7, rng.randint(1, N, E)).astype(np.int32)
        dst = rng.randint(1, N, E).astype(np.int32)
        neg = rng.randint(1, N, E).astype(np.int32)
        ts = np.cumsum(rng.rand(E) * 10.0)
        eidx = np.arange(1, E + 1, dtype=np.int64)
        f = zt.tppr_finder(N, k, 2, [0.1, 0.1], [0.5, 0.95])
        o = oracle.TpprOracle(N, k, 2, [0.1, 0.1], [0.5, 0.95])
        f.chain_stats()
        for s in range(0, E, bs):
            nodes = np.concatenate([src[s:s + bs], dst[s:s + bs], neg[s:s + bs]])
            a = f.streaming_topk(nodes, ts[s:s + bs], eidx[s:s + bs])
            b = o.streaming_topk(nodes, ts[s:s + bs], eidx[s:s + bs])
            for x, y in zip(a, b):
                assert np.array_equal(np.stack(x), np.stack(y))
        st = f.chain_stats()
        assert st["pairs_done"] > 0, st
        for m in range(2):
            _cmp_state(f.export_state(m), o.export(m), "model %d" % m)
    finally:
        _capi.set_kernel_choice(_capi.CHOICE_TPPR_CHAIN, 0)


def test_streaming_variants_golden(zt):
    kind, N, E, seed, bs, k, al, be, _ = I.STREAM_CASES["tiny_general"]
    g = golden("g1_variants")
    src, dst, neg, ts, eidx = I.make_stream(kind, N, E, seed)
    f = zt.tppr_finder(N, k, len(al), al, be)
    on, oe, od, ow = f.streaming_topk_no_fake(np.concatenate([src[:40], dst[:40]]), np.concatenate([ts[:40]] * 2),
                                              eidx[:40])
    assert np.array_equal(np.stack(on), g["nf_nodes"]) and np.array_equal(np.stack(oe), g["nf_eidx"])
    assert np.array_equal(np.stack(od), g["nf_dt"]) and np.array_equal(np.stack(ow), g["nf_w"])
    sn, se, sd, sw = f.single_streaming_topk(np.concatenate([src[40:80], dst[40:80], neg[40:80]]),
                                             np.concatenate([ts[40:80]] * 3), eidx[40:80], 1)
    assert np.array_equal(sn, g["single_nodes"]) and np.array_equal(se, g["single_eidx"])
    assert np.array_equal(sd, g["single_dt"]) and np.array_equal(sw, g["single_w"])
    for m in range(len(al)):
        st = f.export_state(m)
        _cmp_state(st, {kk: g["state%d_%s" % (m, kk)] for kk in st})


def test_fill_and_snapshots(zt):
    kind, N, E, seed, bs, k, al, be, _ = I.STREAM_CASES["bip_k20"]
    g = golden("g2_fill")
    src, dst, neg, ts, eidx = I.make_stream(kind, N, E, seed)
    f = zt.tppr_finder(N, k, len(al), al, be)
    f.compute_val_tppr(src[:1000], dst[:1000], ts[:1000], eidx[:1000], chunk=300)
    for m in range(len(al)):
        st = f.export_state(m)
        _cmp_state(st, {kk: g["state%d_%s" % (m, kk)] for kk in st})
    # deep snapshot semantics (default): backup -> stream -> restore really restores
    before = f.export_state(0)
    bk = f.backup_tppr()
    nodes = np.concatenate([src[1000:1200], dst[1000:1200], neg[1000:1200]])
    f.streaming_topk(nodes, ts[1000:1200], eidx[1000:1200])
    assert not np.array_equal(f.export_state(0)["w"], before["w"])
    f.restore_tppr(bk)
    _cmp_state(f.export_state(0), before)
    f.reset_tppr()
    assert f.export_state(0)["len"].sum() == 0
    f.restore_val_tppr()
    _cmp_state(f.export_state(0), before)
    # reference-compatible aliasing: restore is a no-op, like the reference (g2 records it)
    f2 = zt.tppr_finder(N, k, len(al), al, be, reference_compat_aliasing=True)
    f2.compute_val_tppr(src[:1000], dst[:1000], ts[:1000], eidx[:1000])
    bk = f2.backup_tppr()
    f2.streaming_topk(nodes, ts[1000:1200], eidx[1000:1200])
    after = f2.export_state(0)
    f2.restore_tppr(bk)
    _cmp_state(f2.export_state(0), after)
    assert bool(g["restore_is_noop"])


@pytest.mark.parametrize("cfg", [
    dict(N=2000, E=20000, bs=600, k=20, al=[0.1, 0.1], be=[0.5, 0.95], kind="bipartite", seed=101),
    dict(N=500, E=12000, bs=4096, k=20, al=[0.1, 0.1], be=[0.5, 0.95], kind="hub", seed=102),
    dict(N=3000, E=9000, bs=1000, k=40, al=[0.2], be=[0.8], kind="general", seed=103),
    dict(N=300, E=3000, bs=200, k=63, al=[0.0], be=[0.6], kind="general", seed=104),
    dict(N=400, E=12288, bs=4096, k=30, al=[0.2, 0.0], be=[0.95, 0.5], kind="hub", seed=105),   # the largest k with hub chains
    dict(N=400, E=12288, bs=4096, k=31, al=[0.2, 0.0], be=[0.95, 0.5], kind="hub", seed=106),   # ... and the first without
])
def test_streaming_vs_oracle(zt, oracle, cfg):
    _streaming_vs_oracle(zt, oracle, cfg)


@pytest.mark.parametrize("cfg", [
    dict(N=300, E=3000, bs=200, k=64, al=[0.1, 0.0], be=[0.5, 0.95], kind="hub", seed=201),       # the first k beyond a wavefront; exact ties
    dict(N=60, E=2400, bs=600, k=100, al=[0.2], be=[0.5], kind="general", seed=202),               # dense: rows fill, 2k + 1 = 201 candidates
    dict(N=40, E=3000, bs=1000, k=255, al=[0.1, 0.1], be=[0.25, 0.8], kind="general", seed=203),   # the widest: up to 511 candidates
    dict(N=500, E=1500, bs=500, k=80, al=[0.1], be=[0.9], kind="bipartite", seed=204),             # sparse: rows never fill
])
def test_streaming_wide_k_vs_oracle(zt, oracle, cfg):
    """k beyond ZT_MAX_K (the reference's --topk is unbounded, train.py:46): the one-wavefront-per-model path
    (csrc/tppr_wide.hpp) against the oracle -- every batch's four output arrays and the final state bit-identical; beta = 0.5 /
    0.25 scale exactly (ties decide the top k: numba's argsort order over up to 2k + 1 candidates), alpha = 0, self-loops and
    negatives equal to endpoints come with the stream kinds.  Then the finder's other entry points on the same handle:
    no_fake, single model, backup / restore, export -> import."""
    _streaming_vs_oracle(zt, oracle, cfg)
    src, dst, neg, ts, eidx = I.make_stream(cfg["kind"], cfg["N"], cfg["E"], cfg["seed"])
    al, be, k, N = cfg["al"], cfg["be"], cfg["k"], cfg["N"]
    f = zt.tppr_finder(N, k, len(al), al, be)
    o = oracle.TpprOracle(N, k, len(al), al, be)
    n1 = min(400, cfg["E"] // 2)
    two = np.concatenate([src[:n1], dst[:n1]])
    a, b = f.streaming_topk_no_fake(two, ts[:n1], eidx[:n1]), o.streaming_topk_no_fake(two, ts[:n1], eidx[:n1])
    for x, y in zip(a, b):
        assert np.array_equal(np.stack(x), np.stack(y))
    bk = f.backup_tppr()
    three = np.concatenate([src[n1:2 * n1], dst[n1:2 * n1], neg[n1:2 * n1]])
    a = f.single_streaming_topk(three, ts[n1:2 * n1], eidx[n1:2 * n1], 0)
    b = o.single_streaming_topk(three, ts[n1:2 * n1], eidx[n1:2 * n1], 0)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    for m in range(len(al)):
        _cmp_state(f.export_state(m), o.export(m), "model %d" % m)
    st = f.state_dict(np.arange(N))
    f2 = zt.tppr_finder(N, k, len(al), al, be)
    f2.load_state_dict(st)
    for m in range(len(al)):
        _cmp_state(f2.export_state(m), o.export(m), "model %d after import" % m)
    f.restore_tppr(bk)
    a = f.single_streaming_topk(three, ts[n1:2 * n1], eidx[n1:2 * n1], 0)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    with pytest.raises(IndexError):                                # a rejected launch: state untouched, reported
        f.streaming_topk(np.array([1, 2, N + 3], np.int32), np.array([1.0]), np.array([1], np.int64))
    for m in range(len(al)):
        _cmp_state(f.export_state(m), o.export(m), "model %d after a rejected call" % m)


def _streaming_vs_oracle(zt, oracle, cfg):
    """Seeded streams at sizes the oracle finishes in seconds: every batch's
    four output arrays and the final state must be bit-identical."""
    src, dst, neg, ts, eidx = I.make_stream(cfg["kind"], cfg["N"], cfg["E"], cfg["seed"])
    al, be, k, N, bs, E = cfg["al"], cfg["be"], cfg["k"], cfg["N"], cfg["bs"], cfg["E"]
    f = zt.tppr_finder(N, k, len(al), al, be)
    o = oracle.TpprOracle(N, k, len(al), al, be)
    for s in range(0, E, bs):
        e = min(E, s + bs)
        nodes = np.concatenate([src[s:e], dst[s:e], neg[s:e]])
        a = f.streaming_topk(nodes, ts[s:e], eidx[s:e])
        b = o.streaming_topk(nodes, ts[s:e], eidx[s:e])
        for x, y, nm in zip(a, b, ("nodes", "eidx", "dt", "w")):
            assert np.array_equal(np.stack(x), np.stack(y)), "%s differs in batch at %d" % (nm, s)
    for m in range(len(al)):
        _cmp_state(f.export_state(m), o.export(m), "model %d" % m)


def test_streaming_edge_cases(zt, oracle):
    f = zt.tppr_finder(20, 5, 1, [0.1], [0.9])
    # empty batch
    on, oe, od, ow = f.streaming_topk(np.zeros(0, np.int32), np.zeros(0), np.zeros(0, np.int64))
    assert on[0].shape == (0, 5)
    # out-of-range node id: rejected, state untouched
    with pytest.raises(IndexError):
        f.streaming_topk(np.array([1, 2, 25], np.int32), np.array([1.0]), np.array([1], np.int64))
    assert f.export_state(0)["len"].sum() == 0
    with pytest.raises(IndexError):
        f.streaming_topk(np.array([1, 2, 3], np.int32), np.array([1.0]), np.array([-4], np.int64))
    # a valid call afterwards works and matches the oracle
    o = oracle.TpprOracle(20, 5, 1, [0.1], [0.9])
    nodes = np.array([1, 1, 2, 2, 1, 3, 4, 5, 6], np.int32)        # self-loop, repeated pair
    t = np.array([1.0, 1.0, 2.0])
    e = np.array([1, 2, 3], np.int64)
    a, b = f.streaming_topk(nodes, t, e), o.streaming_topk(nodes, t, e)
    for x, y in zip(a, b):
        assert np.array_equal(np.stack(x), np.stack(y))
    with pytest.raises(ValueError):
        zt.tppr_finder(20, 256, 1, [0.1], [0.9])                   # k > ZT_MAX_K_WIDE


@pytest.mark.parametrize("name", list(I.PRUNE_CASES))
def test_pruning_golden(zt, name):
    kind, N, E, seed, nq, width, depth, k, alpha, beta = I.PRUNE_CASES[name]
    g = golden("g3_prune_" + name)
    src, dst, neg, ts, eidx = I.make_stream(kind, N, E, seed)
    nf = zt.get_neighbor_finder(types.SimpleNamespace(sources=src, destinations=dst, edge_idxs=eidx, timestamps=ts))
    for v in g["probe"]:
        nb, ei, tt = nf.find_before(int(v), np.inf)
        assert np.array_equal(nb, g["adj%d_nbr" % v]) and np.array_equal(ei, g["adj%d_eid" % v])
        assert np.array_equal(tt, g["adj%d_ts" % v])
    on = np.zeros((nq, k), np.int32)
    oe = np.zeros((nq, k), np.int32)
    od = np.zeros((nq, k), np.float32)
    ow = np.zeros((nq, k), np.float32)
    assert nf.get_pruned_topk(g["q_nodes"], g["q_ts"], width, depth, alpha, beta, k, on, oe, od, ow) is None
    assert np.array_equal(on, g["nodes"]) and np.array_equal(oe, g["eidx"])
    assert np.array_equal(od, g["dt"]) and np.array_equal(ow, g["w"])
    # the list-of-arrays constructor gives the same object
    nf2 = zt.NeighborFinder(nf.node_to_neighbors, nf.node_to_edge_idxs, nf.node_to_edge_timestamps)
    on2 = np.full((nq, k), 7, np.int32)
    oe2, od2, ow2 = np.zeros_like(oe), np.zeros_like(od), np.zeros_like(ow)
    nf2.get_pruned_topk(g["q_nodes"], g["q_ts"], width, depth, alpha, beta, k, on2, oe2, od2, ow2)
    empty = (g["w"] == 0).all(axis=1) & (g["nodes"] == 0).all(axis=1) & (g["dt"] == 0).all(axis=1)
    assert np.array_equal(on2[~empty], g["nodes"][~empty])
    assert (on2[empty] == 7).all()                    # rows with an empty dictionary are left untouched


def test_pruning_vs_oracle_superuser_shape(zt, oracle):
    """C4-shaped (k=40, width 10, depth 2) at a size the oracle runs in seconds."""
    N, E = 20000, 150000
    src, dst, neg, ts, eidx = I.make_stream("general", N, E, 105)
    nf = zt.get_neighbor_finder(types.SimpleNamespace(sources=src, destinations=dst, edge_idxs=eidx, timestamps=ts))
    csr = oracle.CsrOracle(src, dst, eidx, ts, nf.num_nodes)
    s = E - 1000
    q = np.concatenate([src[s:], dst[s:], neg[s:]])
    qt = np.concatenate([ts[s:]] * 3)
    for alpha, beta in ((0.1, 0.5), (0.1, 0.95)):
        outs_a = [np.zeros((3000, 40), dt) for dt in (np.int32, np.int32, np.float32, np.float32)]
        outs_b = [np.zeros((3000, 40), dt) for dt in (np.int32, np.int32, np.float32, np.float32)]
        nf.get_pruned_topk(q, qt, 10, 2, alpha, beta, 40, *outs_a)
        csr.get_pruned_topk(q, qt, 10, 2, alpha, beta, 40, *outs_b)
        for x, y in zip(outs_a, outs_b):
            assert np.array_equal(x, y)
    with pytest.raises(ValueError):
        nf.get_pruned_topk(q[:4], qt[:4], 20, 3, 0.1, 0.5, 20, *[np.zeros((4, 20), dt) for dt in
                                                                 (np.int32, np.int32, np.float32, np.float32)])


@pytest.mark.parametrize("k,width,depth,beta", [(64, 12, 2, 0.5), (100, 20, 2, 0.5), (255, 30, 2, 0.8), (80, 6, 3, 0.95)])
def test_pruning_wide_k_vs_oracle(zt, oracle, k, width, depth, beta):
    """get_pruned_topk with k beyond a wavefront (the reference puts no bound on --topk, train.py:46): the kept set strides
    over the lanes and the selection is the generic replay; beta = 0.5 makes exact ties at the cut.  Bit-identical to the
    oracle, rows with fewer than k candidates included."""
    N, E = 400, 24000
    src, dst, neg, ts, eidx = I.make_stream("hub", N, E, 305)
    nf = zt.get_neighbor_finder(types.SimpleNamespace(sources=src, destinations=dst, edge_idxs=eidx, timestamps=ts))
    csr = oracle.CsrOracle(src, dst, eidx, ts, nf.num_nodes)
    s = E - 300
    q = np.concatenate([src[s:], dst[s:], neg[s:]])
    qt = np.concatenate([ts[s:]] * 3)
    outs_a = [np.zeros((900, k), dt) for dt in (np.int32, np.int32, np.float32, np.float32)]
    outs_b = [np.zeros((900, k), dt) for dt in (np.int32, np.int32, np.float32, np.float32)]
    nf.get_pruned_topk(q, qt, width, depth, 0.1, beta, k, *outs_a)
    csr.get_pruned_topk(q, qt, width, depth, 0.1, beta, k, *outs_b)
    for x, y, nm in zip(outs_a, outs_b, ("nodes", "eidx", "dt", "w")):
        assert np.array_equal(x, y), nm
    assert (outs_b[3][:, 0] != 0).any() and (np.count_nonzero(outs_b[3], axis=1) == k).any()     # some rows are full


@pytest.mark.parametrize("n,k", [(16, 5), (17, 5), (31, 20), (41, 20), (41, 40), (63, 31), (64, 20), (65, 31), (81, 40),
                                 (100, 50), (127, 63), (128, 20), (11, 5), (21, 10)])
def test_exact_topk_selection_paths(zt, oracle, n, k):
    """The top-k prune must reproduce np.argsort(values)[-k:] under numba's
    quicksort for any tie pattern.  Every path (rank fast path, wave-parallel
    quicksort replay, sequential replay) is driven on its own with tie-heavy
    inputs and compared with the oracle's restatement."""
    import ctypes as C
    import torch
    from zebra_amd import _capi
    rng = np.random.RandomState(n * 100 + k)
    cases = 600
    vals = np.empty((cases, n), np.float64)
    for c in range(cases):
        kind = c % 6
        if kind == 0:
            vals[c] = rng.random_sample(n)                                  # all distinct
        elif kind == 1:
            vals[c] = rng.randint(0, 3, n)                                  # massive ties
        elif kind == 2:
            vals[c] = rng.randint(0, max(2, n // 3), n) * 0.125             # some ties
        elif kind == 3:
            vals[c] = np.sort(rng.randint(0, 6, n))[::(-1 if c % 2 else 1)]  # sorted / reversed with ties
        elif kind == 4:
            vals[c] = 0.5 ** rng.randint(1, 8, n)                           # powers of beta=0.5, like the weights
        else:
            vals[c] = 1.0                                                   # all equal
    want = np.stack([oracle.numba_argsort(v)[-k:] for v in vals])
    dv = torch.from_numpy(vals).cuda()
    for mode in (0, 1, 2, 3, 4, 5, 6):
        if mode in (3, 4) and n > 64:
            continue
        if mode in (5, 6) and (n > 63 or k > 31):
            continue
        sel = torch.full((cases, k), -1, dtype=torch.int32, device="cuda")
        path = torch.full((cases,), -1, dtype=torch.int32, device="cuda")
        _capi.check(_capi.hooks_lib().zt_test_topk(_capi.ptr(dv), C.c_int32(n), C.c_int32(k), C.c_int32(cases),
                                             C.c_int32(mode), _capi.ptr(sel), _capi.ptr(path), _capi.stream_ptr()))
        got = sel.cpu().numpy()
        bad = np.where((got != want).any(axis=1))[0]
        assert len(bad) == 0, "mode %d: %d/%d cases differ, first %d: %s vs %s" % (
            mode, len(bad), cases, bad[0] if len(bad) else -1, got[bad[0]] if len(bad) else None,
            want[bad[0]] if len(bad) else None)
        if mode == 0:
            p = path.cpu().numpy() & 0xff
            if n <= 64:
                assert (p == 0).any() and (p == 4).any() and ((p == 0) | (p == 4)).all()   # rank path, replay on ranks
            else:
                assert (p == 0).any() and (p == 5).any() and ((p == 0) | (p == 5)).all()   # ranks, two-position replay


def test_large_single_call_and_epoch_wrap(zt, oracle):
    """One call with more edges than a launch chunk (16384), and launches across
    the tag-epoch wrap-around, still match the oracle bit for bit."""
    import ctypes as C
    from zebra_amd import _capi
    N, E, k = 4000, 30000, 20
    al, be = [0.1, 0.1], [0.5, 0.95]
    src, dst, neg, ts, eidx = I.make_stream("bipartite", N, E, 401)
    f = zt.tppr_finder(N, k, 2, al, be)
    o = oracle.TpprOracle(N, k, 2, al, be)
    # 20000 edges in ONE call -> two device launches
    nodes = np.concatenate([src[:20000], dst[:20000], neg[:20000]])
    a = f.streaming_topk(nodes, ts[:20000], eidx[:20000])
    b = o.streaming_topk(nodes, ts[:20000], eidx[:20000])
    for x, y in zip(a, b):
        assert np.array_equal(np.stack(x), np.stack(y))
    # jump next to the epoch wrap and keep streaming across it
    _capi.check(_capi.hooks_lib().zt_test_set_epoch(f._live.h, C.c_uint32((1 << 17) - 4)))
    for s in range(20000, 30000, 1000):
        e = s + 1000
        nodes = np.concatenate([src[s:e], dst[s:e], neg[s:e]])
        a = f.streaming_topk(nodes, ts[s:e], eidx[s:e])
        b = o.streaming_topk(nodes, ts[s:e], eidx[s:e])
        for x, y in zip(a, b):
            assert np.array_equal(np.stack(x), np.stack(y)), "differs after epoch jump at %d" % s
    for m in range(2):
        _cmp_state(f.export_state(m), o.export(m))


def test_full_size_properties(zt):
    """BASELINE.json's full-size configuration (C5: 10 M nodes, bs=4096, k=20, two models) is far beyond
    what the oracle replays in a test, so the streaming path is checked there through properties that do
    not depend on the size: (1) batch-split invariance -- the reference applies edges one by one
    (utils/util.py:499-574), so a batch applied in one call, in two halves or in ragged pieces must
    leave bit-identical dictionaries, and the rows emitted for an edge only depend on the edges before
    it; (2) determinism under the scheduler -- two runs of the same stream agree bit for bit although
    thousands of wavefronts race through the per-node chains; (3) kept weights and norms are positive,
    keys inside a dictionary are unique, and the newest key of the hub is the last edge that touched it."""
    import torch
    from zebra_amd import synth
    wl = synth.WORKLOADS["c5"]
    N, B, k = wl["n_nodes"], 4096, 20
    al, be = [0.1, 0.1], [0.5, 0.95]
    nb_fill, nb_chk = 60, 4
    src, dst, ts, eidx = synth.power_law_stream(N, (nb_fill + nb_chk) * B, seed=77)
    neg = synth.negatives(dst, len(src), seed=78)
    d = torch.device("cuda")
    sd, dd, nd = [torch.from_numpy(x).to(d) for x in (src, dst, neg)]
    td, ed = torch.from_numpy(ts).to(d), torch.from_numpy(eidx).to(d)

    def run(pieces):
        """pieces(b) -> list of (lo, hi) covering batch b's edges in order"""
        f = zt.tppr_finder(N + 1, k, 2, al, be)
        outs = []
        for b in range(nb_fill + nb_chk):
            rows = [[], [], [], []]
            for lo, hi in (pieces(b) if b >= nb_fill else [(b * B, (b + 1) * B)]):
                o = f.stream_device(torch.cat([sd[lo:hi], dd[lo:hi], nd[lo:hi]]), td[lo:hi], ed[lo:hi], 3, True, -1,
                                    check_status=False)
                n = hi - lo
                for q in range(4):      # [models][3n][k] -> per role [models][n][k]
                    rows[q].append(o[q].reshape(2, 3, n, k))
            if b >= nb_fill:
                outs.append([torch.cat(r, dim=2) for r in rows])
        f.check_status()
        return f, outs

    whole = lambda b: [(b * B, (b + 1) * B)]
    halves = lambda b: [(b * B, b * B + B // 2), (b * B + B // 2, (b + 1) * B)]
    ragged = lambda b: [(b * B, b * B + 1), (b * B + 1, b * B + 700), (b * B + 700, b * B + 701),
                        (b * B + 701, (b + 1) * B)]
    fa, oa = run(whole)
    for name, pc in (("halves", halves), ("ragged", ragged), ("whole again", whole)):
        fb, ob = run(pc)
        for x, y in zip(oa, ob):
            for q in range(4):
                assert torch.equal(x[q], y[q]), "%s: emitted rows differ" % name
        hot = np.unique(np.concatenate([src, dst])).astype(np.int64)
        for m in range(2):
            sa, sb = fa.export_rows(m, hot), fb.export_rows(m, hot)
            for kk in sa:
                assert np.array_equal(sa[kk], sb[kk]), "%s: state %s of model %d differs" % (name, kk, m)
        del fb
    hot = np.unique(np.concatenate([src, dst])).astype(np.int64)
    hub = int(np.bincount(np.concatenate([src, dst])).argmax())
    last = int(np.where((src == hub) | (dst == hub))[0].max())
    for m in range(2):
        st = fa.export_rows(m, hot)
        live = np.arange(k)[None, :] < st["len"][:, None]
        assert (st["len"] >= 1).all() and (st["len"] <= k).all()
        assert (st["w"][live] > 0).all() and (st["norm"] > 0).all()
        key = st["eidx"] * (N + 2) + st["node"]
        key[~live] = -1 - np.arange((~live).sum())            # padding: all different
        srt = np.sort(key, axis=1)
        assert (np.diff(srt, axis=1) != 0).all()
        hrow = fa.export_rows(m, np.array([hub], np.int64))
        assert eidx[last] in hrow["eidx"][0, : hrow["len"][0]]


def test_checkpoint_round_trip(zt, oracle, tmp_path):
    """a run interrupted after some batches, checkpointed through ``state_dict`` (only
    the touched nodes, written with np.savez), restored into a NEW finder and continued must equal the
    uninterrupted run and the oracle bit for bit."""
    N, E, k, bs = 3000, 12000, 20, 600
    al, be = [0.1, 0.1], [0.5, 0.95]
    src, dst, neg, ts, eidx = I.make_stream("general", N, E, 811)
    o = oracle.TpprOracle(N, k, 2, al, be)
    a = zt.tppr_finder(N, k, 2, al, be)
    half = (E // bs // 2) * bs

    def feed(f, s, e):
        nodes = np.concatenate([src[s:e], dst[s:e], neg[s:e]])
        return f.streaming_topk(nodes, ts[s:e], eidx[s:e])

    for s in range(0, half, bs):
        feed(a, s, s + bs)
        feed(o, s, s + bs)
    touched = np.unique(np.concatenate([src[:half], dst[:half]]))
    path = tmp_path / "tppr.npz"
    np.savez(path, **a.state_dict(touched))
    b = zt.tppr_finder(N, k, 2, al, be)
    b.load_state_dict(dict(np.load(path)))
    for m in range(2):
        _cmp_state(b.export_state(m), a.export_state(m), "restored model %d" % m)
    for s in range(half, E, bs):
        ra, rb, ro = feed(a, s, s + bs), feed(b, s, s + bs), feed(o, s, s + bs)
        for x, y, z in zip(ra, rb, ro):
            assert np.array_equal(np.stack(x), np.stack(y)) and np.array_equal(np.stack(y), np.stack(z))
    for m in range(2):
        _cmp_state(b.export_state(m), o.export(m), "continued model %d" % m)
    with pytest.raises(Exception):
        zt.tppr_finder(N, 10, 2, al, be).load_state_dict(dict(np.load(path)))


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_dense_hub_graph_vs_oracle(zt, oracle, seed):
    """Every node is a hub: 40 nodes, 4096-edge launches, so all 16 chain workgroups are in use,
    chains own edges whose other endpoint is another chain's hub (foreign writers between mailbox
    hand-offs), negatives hit hubs (readers between hand-offs: the row must go to memory, the next
    writer must wait for the reads-done flag), plus self-loops and negative == endpoint.  Bit-exact
    against the oracle, outputs and final state."""
    rng = np.random.RandomState(900 + seed)
    N, E, k, bs = 41, 3 * 4096, 20, 4096
    p = 1.0 / np.arange(1, N) ** 0.7
    p /= p.sum()
    src = (1 + rng.choice(N - 1, E, p=p)).astype(np.int32)
    dst = (1 + rng.choice(N - 1, E, p=p)).astype(np.int32)
    loops = rng.random_sample(E) < 0.03
    dst[loops] = src[loops]
    neg = (1 + rng.choice(N - 1, E, p=p)).astype(np.int32)
    same = rng.random_sample(E) < 0.05
    neg[same] = src[same]
    ts = np.cumsum(rng.randint(0, 3, E)).astype(np.float64)          # repeated timestamps too
    eidx = np.arange(1, E + 1, dtype=np.int64)
    al, be = [0.1, 0.2], [0.5, 0.95]
    f = zt.tppr_finder(N, k, 2, al, be)
    o = oracle.TpprOracle(N, k, 2, al, be)
    for s in range(0, E, bs):
        e = s + bs
        nodes = np.concatenate([src[s:e], dst[s:e], neg[s:e]])
        a = f.streaming_topk(nodes, ts[s:e], eidx[s:e])
        b = o.streaming_topk(nodes, ts[s:e], eidx[s:e])
        for x, y, nm in zip(a, b, ("nodes", "eidx", "dt", "w")):
            assert np.array_equal(np.stack(x), np.stack(y)), "%s differs in batch at %d" % (nm, s)
    for m in range(2):
        _cmp_state(f.export_state(m), o.export(m), "model %d" % m)


def test_chain_overflow_vs_oracle(zt, oracle):
    """Six nodes, one 8192-edge launch: every hub is touched by more edges than a chain holds
    (CH_MAX = 2048), so chains hand the rest to the general queue in the middle of their order."""
    rng = np.random.RandomState(77)
    N, E, k = 7, 8192, 5
    src = rng.randint(1, N, E).astype(np.int32)
    dst = rng.randint(1, N, E).astype(np.int32)
    neg = rng.randint(1, N, E).astype(np.int32)
    ts = np.arange(E, dtype=np.float64)
    eidx = np.arange(1, E + 1, dtype=np.int64)
    al, be = [0.15], [0.9]
    f = zt.tppr_finder(N, k, 1, al, be)
    o = oracle.TpprOracle(N, k, 1, al, be)
    nodes = np.concatenate([src, dst, neg])
    a, b = f.streaming_topk(nodes, ts, eidx), o.streaming_topk(nodes, ts, eidx)
    for x, y in zip(a, b):
        assert np.array_equal(np.stack(x), np.stack(y))
    _cmp_state(f.export_state(0), o.export(0))


def test_long_power_law_stream_vs_oracle(zt, oracle):
    """A bench-shaped stream (power-law endpoints, bs=4096, k=20, alpha=[.1,.1], beta=[.5,.95]) long enough
    for the dictionaries of the hubs to be full and tie-laden (beta = 0.5 scales exactly): 300 launches of
    4096 edges on 100,000 nodes, checked against the oracle every 20th batch (all four output arrays) and on
    the final state of every touched node."""
    from zebra_amd import synth
    N, B, k, nb = 100_000, 4096, 20, 300
    al, be = [0.1, 0.1], [0.5, 0.95]
    src, dst, ts, eidx = synth.power_law_stream(N, nb * B, seed=515)
    neg = synth.negatives(dst, len(src), seed=516)
    f = zt.tppr_finder(N + 1, k, 2, al, be)
    o = oracle.TpprOracle(N + 1, k, 2, al, be)
    for b in range(nb):
        s, e = b * B, (b + 1) * B
        nodes = np.concatenate([src[s:e], dst[s:e], neg[s:e]])
        ob = o.streaming_topk(nodes, ts[s:e], eidx[s:e])
        if b % 20 == 19 or b == nb - 1:
            a = f.streaming_topk(nodes, ts[s:e], eidx[s:e])
            for x, y, nm in zip(a, ob, ("nodes", "eidx", "dt", "w")):
                assert np.array_equal(np.stack(x), np.stack(y)), "%s differs in batch %d" % (nm, b)
        else:
            import torch
            d = torch.device("cuda")
            f.stream_device(torch.from_numpy(nodes).to(d), torch.from_numpy(ts[s:e]).to(d),
                            torch.from_numpy(eidx[s:e]).to(d), 3, True, -1, check_status=False)
    f.check_status()
    touched = np.unique(np.concatenate([src, dst])).astype(np.int64)
    for m in range(2):
        got, want = f.export_rows(m, touched), o.export(m)
        for kk in got:
            assert np.array_equal(got[kk], want[kk][touched]), "state %s of model %d" % (kk, m)


# ---------------------------------------------------------------------------------------------------------
# failure latch, plan tokens, grids that do not fit the stream (round-1 advisor findings)
# ---------------------------------------------------------------------------------------------------------
def _dev_batch(torch, src, dst, neg, ts, eidx, s, e):
    d = torch.device("cuda")
    return (torch.from_numpy(np.concatenate([src[s:e], dst[s:e], neg[s:e]])).to(d), torch.from_numpy(ts[s:e]).to(d),
            torch.from_numpy(eidx[s:e]).to(d))


def test_unchecked_bad_batch_is_dropped_and_latched(zt, oracle):
    """A batch with an out-of-range id run WITHOUT a status check: its rows read as empty dictionaries, the state
    is untouched, every valid batch that still gets through matches the oracle, and a later call raises."""
    import torch
    N, E, bs, k = 400, 4000, 200, 20
    al, be = [0.1, 0.1], [0.5, 0.95]
    src, dst, neg, ts, eidx = I.make_stream("bipartite", N, E, 111)
    f = zt.tppr_finder(N, k, 2, al, be)
    o = oracle.TpprOracle(N, k, 2, al, be)
    raised = False
    for b, s in enumerate(range(0, E, bs)):
        e = s + bs
        nodes_d, ts_d, eidx_d = _dev_batch(torch, src, dst, neg, ts, eidx, s, e)
        bad = b == 3
        if bad:
            nodes_d = nodes_d.clone()
            nodes_d[17] = N + 5
        try:
            outs = f.stream_device(nodes_d, ts_d, eidx_d, 3, True, -1, check_status=False)
        except IndexError:
            raised = True
            assert b > 3
            break
        got = [x.cpu().numpy() for x in outs]
        if bad:
            for x in got:
                assert not x.any(), "rows of a rejected batch must be zero"
            continue
        want = o.streaming_topk(np.concatenate([src[s:e], dst[s:e], neg[s:e]]), ts[s:e], eidx[s:e])
        for x, y in zip(got, want):
            assert np.array_equal(x, np.stack(y)), "batch %d after the rejected one" % b
    if not raised:
        with pytest.raises(IndexError):
            f.check_status()
    # reported once; the handle works again and still equals the oracle (which never saw the bad batch)
    f.check_status()
    nodes = np.concatenate([src[:50], dst[:50], neg[:50]])
    a, w = f.streaming_topk(nodes, ts[:50] + ts[-1], eidx[:50]), o.streaming_topk(nodes, ts[:50] + ts[-1], eidx[:50])
    for x, y in zip(a, w):
        assert np.array_equal(np.stack(x), np.stack(y))
    for m in range(2):
        _cmp_state(f.export_state(m), o.export(m))


def test_plan_tokens(zt, oracle):
    """A plan is used only by the call that presents its token; a token whose plan has been dropped
    (reset) falls back to an inline prepass; a token presented with other arguments is an error."""
    import torch
    N, E, bs, k = 300, 1200, 300, 20
    src, dst, neg, ts, eidx = I.make_stream("hub", N, E, 112)
    f = zt.tppr_finder(N, k, 1, [0.1], [0.5])
    o = oracle.TpprOracle(N, k, 1, [0.1], [0.5])
    b0, b1 = _dev_batch(torch, src, dst, neg, ts, eidx, 0, bs), _dev_batch(torch, src, dst, neg, ts, eidx, bs, 2 * bs)
    tok0 = f.plan_device(b0[0], b0[2], 3, -1)
    tok1 = f.plan_device(b1[0], b1[2], 3, -1)
    assert tok0 != 0 and tok1 != 0 and tok0 != tok1
    with pytest.raises(ValueError):                       # batch 1's ids with batch 0's token
        f.stream_device(b1[0], b1[1], b1[2], 3, True, -1, plan_token=tok0)
    for (n_d, t_d, e_d), tok, s in ((b0, tok0, 0), (b1, tok1, bs)):
        got = f.stream_device(n_d, t_d, e_d, 3, True, -1, plan_token=tok)
        want = o.streaming_topk(np.concatenate([src[s:s + bs], dst[s:s + bs], neg[s:s + bs]]), ts[s:s + bs], eidx[s:s + bs])
        for x, y in zip(got, want):
            assert np.array_equal(x.cpu().numpy(), np.stack(y))
    # a plan made before reset_tppr must not be applied after it; same address, different ids
    tok = f.plan_device(b0[0], b0[2], 3, -1)
    f.reset_tppr()
    o.reset_tppr()
    b0[0].copy_(b1[0])
    got = f.stream_device(b0[0], b1[1], b1[2], 3, True, -1, plan_token=tok)
    want = o.streaming_topk(np.concatenate([src[bs:2 * bs], dst[bs:2 * bs], neg[bs:2 * bs]]), ts[bs:2 * bs], eidx[bs:2 * bs])
    for x, y in zip(got, want):
        assert np.array_equal(x.cpu().numpy(), np.stack(y))
    _cmp_state(f.export_state(0), o.export(0))


def test_grid_larger_than_the_run_stream(zt, oracle):
    """Planned for the whole chip, run on a CU-masked stream of 8 CUs: the grid must shrink to what is resident
    there and the hub chains must give way to the in-order queue -- same results, no time-out."""
    import ctypes as C
    import torch
    from zebra_amd._capi import lib, check
    N, E, bs, k = 500, 8192, 4096, 20
    al, be = [0.1, 0.1], [0.5, 0.95]
    src, dst, neg, ts, eidx = I.make_stream("hub", N, E, 113)
    f = zt.tppr_finder(N, k, 2, al, be)
    o = oracle.TpprOracle(N, k, 2, al, be)
    hs = C.c_void_p()
    check(lib().zt_stream_create_masked(C.byref(hs), C.c_int32(0), C.c_int32(8)))
    small = torch.cuda.ExternalStream(hs.value)
    try:
        for s in range(0, E, bs):
            e = s + bs
            n_d, t_d, e_d = _dev_batch(torch, src, dst, neg, ts, eidx, s, e)
            tok = f.plan_device(n_d, e_d, 3, -1)            # on the default stream: plans a 256-CU grid
            torch.cuda.synchronize()
            with torch.cuda.stream(small):
                got = f.stream_device(n_d, t_d, e_d, 3, True, -1, plan_token=tok)
            small.synchronize()
            want = o.streaming_topk(np.concatenate([src[s:e], dst[s:e], neg[s:e]]), ts[s:e], eidx[s:e])
            for x, y in zip(got, want):
                assert np.array_equal(x.cpu().numpy(), np.stack(y))
        for m in range(2):
            _cmp_state(f.export_state(m), o.export(m))
    finally:
        torch.cuda.synchronize()
        check(lib().zt_stream_destroy(hs))


def test_randomised_soak(zt, oracle):
    """A slice of tests/soak_tppr.py (random graph shapes, k, batch sizes up to 16384-edge launches, alpha / beta with
    exact ties, self-loops, negatives on endpoints and hubs, repeated timestamps): every batch and the final state
    bit-identical to the oracle.  Seed 1011 (k = 31, full partner rows) is the one that found the lane-63 bug."""
    import soak_tppr
    for seed in list(range(1000, 1050)) + [1011]:
        err = soak_tppr.one(seed, zt, oracle)
        assert err is None, err


def test_randomised_prune_soak(zt, oracle):
    """A slice of tests/soak_prune.py: random static graphs, width, depth, k (candidate lists from a few to > 128 entries),
    alpha / beta with exact ties, ties in time, nodes without history; single-model and multi-model launches of
    k_pruned_topk bit-identical to the oracle."""
    import torch
    import soak_prune
    for seed in range(5000, 5200):
        err = soak_prune.one(seed, zt, oracle, torch)
        assert err is None, err


@pytest.mark.parametrize("name,batch", [("bip_k20", 0), ("bip_k20", 4), ("hub_ties", 0), ("gen_k40", 1)])
def test_dependency_plan_matches_restatement(zt, name, batch):
    """The prepass of a launch (csrc/tppr_prepass.hip; the order utils/util.py:495-574 applies a batch's edges in, made
    explicit) against a numpy restatement, through the test hook zt_test_tppr_plan_dump: writer ordinals, chain positions
    (= the hub's ordinal; an edge between two hubs in both chains), owners, and which accesses read a row by version."""
    import ctypes as C
    import torch
    from zebra_amd import _capi
    kind, N, E, seed, bs, k, al, be, _ = I.STREAM_CASES[name]
    src, dst, neg, ts, eidx = I.make_stream(kind, N, E, seed)
    f = zt.tppr_finder(N, k, len(al), al, be)
    s0, s1 = batch * bs, min(E, (batch + 1) * bs)
    B = s1 - s0
    nodes = np.concatenate([src[s0:s1], dst[s0:s1], neg[s0:s1]]).astype(np.int32)
    nd = torch.from_numpy(nodes).cuda()
    ed = torch.from_numpy(eidx[s0:s1].astype(np.int64)).cuda()
    tok = f.plan_device(nd, ed, 3, -1)
    assert tok != 0
    wo = np.zeros(3 * B, np.int32); pf = np.zeros_like(wo); hv = np.zeros_like(wo); own = np.zeros(B, np.int32)
    cn = np.zeros(16, np.int32); cl = np.zeros(16, np.int32); ce = np.zeros((16, 2048), np.int32); nc = np.zeros(1, np.int32)
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    _capi.check(_capi.hooks_lib().zt_test_tppr_plan_dump(f._live.h, P(wo), P(pf), P(hv), P(own), P(cn), P(cl), P(ce), P(nc)))
    n_ch = int(nc[0])
    u, v, g = nodes[:B], nodes[B:2 * B], nodes[2 * B:]
    shadow = lambda r, i: (r >= 1 and nodes[r * B + i] == u[i]) or (r == 2 and g[i] == v[i])
    cnt, wr, last = {}, {}, {}
    wo_ref = np.zeros(3 * B, np.int32); pf_ref = np.full(3 * B, -1, np.int32)
    for i in range(B):
        for r in range(3):
            x = int(nodes[r * B + i])
            if shadow(r, i):
                continue
            cnt[x] = cnt.get(x, 0) + 1
            wo_ref[r * B + i] = wr.get(x, 0)
            if x in last and last[x][1] == 2:
                pf_ref[r * B + i] = last[x][0]
        for r in range(3):                                      # (an edge's accesses see the edges before it only)
            if not shadow(r, i):
                x = int(nodes[r * B + i])
                if r < 2:
                    wr[x] = wr.get(x, 0) + 1
        for r in (2, 1, 0):                                     # the node's latest access: a writer role wins over the reader's
            if not shadow(r, i):
                last[int(nodes[r * B + i])] = (i, r)
    assert np.array_equal(wo, wo_ref) and np.array_equal(pf, pf_ref)
    hot = sorted([x for x in cnt if cnt[x] >= 24], key=lambda x: (-cnt[x], x))
    assert n_ch <= len(hot) and list(cn[:n_ch]) == hot[:n_ch] and (n_ch > 0 or name != "bip_k20")
    chain_of = {int(cn[c]): c for c in range(n_ch)}
    for c in range(n_ch):
        edges = [i for i in range(B) if u[i] == cn[c] or v[i] == cn[c]]
        assert cl[c] == len(edges) and list(ce[c, :cl[c]]) == edges, "chain %d" % c
    hv_ref = np.full(3 * B, -1, np.int32); own_ref = np.full(B, -1, np.int32)
    for i in range(B):
        for r in range(3):
            x = int(nodes[r * B + i])
            if not shadow(r, i) and x in chain_of and cl[chain_of[x]] > 0:
                hv_ref[r * B + i] = chain_of[x]
        a, b = int(u[i]), int(v[i])
        ia, ib = a in chain_of, (b in chain_of and b != a)
        if ia and (not ib or cnt[a] >= cnt[b]):
            own_ref[i] = chain_of[a]
        elif ib:
            own_ref[i] = chain_of[b]
    assert np.array_equal(hv, hv_ref) and np.array_equal(own, own_ref)
    f.stream_device(nd, torch.from_numpy(ts[s0:s1].astype(np.float64)).cuda(), ed, 3, True, -1, plan_token=tok)


@pytest.mark.parametrize("shape", ["many_big_groups", "one_huge_group", "big_list_overflow"])
def test_dependency_plan_cooperative_kernel(zt, oracle, shape):
    """The same plans from k_prepass_coop (ZT_PREPASS_COOP: the ten steps in ONE kernel of 24 workgroups, grid barriers
    without fences -- the steps hand their arrays over through write-through stores and sc1 loads), and the goldens' state
    after streaming with it."""
    from zebra_amd import _capi
    _variant_choice(_capi.CHOICE_TPPR_PREPASS, _capi.PREPASS_COOP)
    try:
        test_dependency_plan_of_large_launches(zt, shape)
        if shape == "many_big_groups":
            test_large_single_call_and_epoch_wrap(zt, oracle)
    finally:
        _capi.set_kernel_choice(_capi.CHOICE_TPPR_PREPASS, 0)


@pytest.mark.parametrize("shape", ["many_big_groups", "one_huge_group", "big_list_overflow", "small_fused"])
def test_dependency_plan_of_large_launches(zt, shape):
    """Writer ordinals and reader flags of launches that take every branch of the prepass' dependency step
    (csrc/tppr_prepass.hip: d_deps per access, d_deps_group = a cooperative sort per group of >= 48 accesses, the
    per-access loop again for groups beyond 4096 members or when more than 512 groups qualify; the single-workgroup form
    up to 4096 accesses, ten launches beyond), against the order utils/util.py:495-574 applies a batch's edges in."""
    import ctypes as C
    import torch
    from zebra_amd import _capi
    rng = np.random.RandomState(17)
    if shape == "many_big_groups":
        B, N = 8192, 3000
        p = np.arange(1, N, dtype=np.float64) ** -1.0
        draw = lambda n: 1 + rng.choice(N - 1, n, p=p / p.sum())
        u, v, g = draw(B), draw(B), draw(B)
    elif shape == "one_huge_group":
        B, N = 16384, 5000
        u = np.where(rng.random_sample(B) < 0.55, 7, 1 + rng.randint(0, N - 1, B))
        v, g = 1 + rng.randint(0, N - 1, B), 1 + rng.randint(0, N - 1, B)
    elif shape == "big_list_overflow":
        B, N = 16384, 601
        u, v, g = (1 + rng.randint(0, N - 1, B) for _ in range(3))
    else:
        B, N = 1200, 300
        p = np.arange(1, N, dtype=np.float64) ** -1.2
        draw = lambda n: 1 + rng.choice(N - 1, n, p=p / p.sum())
        u, v, g = draw(B), draw(B), draw(B)
    nodes = np.concatenate([u, v, g]).astype(np.int32)
    f = zt.tppr_finder(N, 20, 2, [0.1, 0.1], [0.5, 0.95])
    nd = torch.from_numpy(nodes).cuda()
    ed = torch.arange(1, B + 1, dtype=torch.int64, device="cuda")
    tok = f.plan_device(nd, ed, 3, -1)
    assert tok != 0
    wo = np.zeros(3 * B, np.int32); pf = np.zeros_like(wo); hv = np.zeros_like(wo); own = np.zeros(B, np.int32)
    cn = np.zeros(16, np.int32); cl = np.zeros(16, np.int32); ce = np.zeros((16, 2048), np.int32); nc = np.zeros(1, np.int32)
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    _capi.check(_capi.hooks_lib().zt_test_tppr_plan_dump(f._live.h, P(wo), P(pf), P(hv), P(own), P(cn), P(cl), P(ce), P(nc)))
    wr, last, cnt = {}, {}, {}
    wo_ref = np.zeros(3 * B, np.int32); pf_ref = np.full(3 * B, -1, np.int32)
    for i in range(B):
        acc = []
        for r in range(3):
            x = int(nodes[r * B + i])
            if (r >= 1 and x == int(u[i])) or (r == 2 and x == int(v[i])):
                continue
            acc.append((r, x))
            cnt[x] = cnt.get(x, 0) + 1
            wo_ref[r * B + i] = wr.get(x, 0)
            if x in last and last[x][1] == 2:
                pf_ref[r * B + i] = last[x][0]
        for r, x in acc:
            if r < 2:
                wr[x] = wr.get(x, 0) + 1
        for r, x in reversed(acc):
            last[x] = (i, r)
    big = [c for c in cnt.values() if c >= 48]
    if shape == "many_big_groups":
        assert 20 < len(big) <= 512 and max(big) <= 4096
    elif shape == "one_huge_group":
        assert max(big) > 4096
    elif shape == "big_list_overflow":
        assert len(big) > 512
    assert np.array_equal(wo, wo_ref), "writer ordinals"
    assert np.array_equal(pf, pf_ref), "reader flags"
    ts = torch.arange(1, B + 1, dtype=torch.float64, device="cuda")
    f.stream_device(nd, ts, ed, 3, True, -1, plan_token=tok)
