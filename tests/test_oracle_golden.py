"""Pins the CPU oracle to the golden vectors produced by the reference itself
(tests/golden/gen_golden.py).  Bit-exact for T-PPR arrays and state; 1e-4 for
float32 network outputs (BASELINE.json north_star)."""
import numpy as np
import pytest

import inputs as I
from helpers import epoch_protocol, make_checker
from conftest import golden

EMB_TOL = 1e-4


def _state_equal(oracle_state, g, m):
    for kk in ("len", "norm", "eidx", "node", "ts", "w"):
        a, b = oracle_state[kk], g["state%d_%s" % (m, kk)]
        assert a.shape == b.shape and np.array_equal(a, b), "state mismatch %s model %d" % (kk, m)


@pytest.mark.parametrize("name", list(I.STREAM_CASES))
def test_streaming_matches_reference(oracle, name):
    kind, N, E, seed, bs, k, al, be, _ = I.STREAM_CASES[name]
    g = golden("g1_stream_" + name)
    src, dst, neg, ts, eidx = I.make_stream(kind, N, E, seed)
    f = oracle.TpprOracle(N, k, len(al), al, be)
    nb = (E + bs - 1) // bs
    full = set(g["full_idx"].tolist())
    for b in range(nb):
        s, e = b * bs, min(E, (b + 1) * bs)
        nodes = np.concatenate([src[s:e], dst[s:e], neg[s:e]])
        on, oe, od, ow = f.streaming_topk(nodes, np.concatenate([ts[s:e]] * 3), eidx[s:e])
        arrs = []
        for m in range(len(al)):
            arrs += [on[m], oe[m], od[m], ow[m]]
        if b in full:
            assert np.array_equal(np.stack(on), g["b%d_nodes" % b])
            assert np.array_equal(np.stack(oe), g["b%d_eidx" % b])
            assert np.array_equal(np.stack(od), g["b%d_dt" % b])
            assert np.array_equal(np.stack(ow), g["b%d_w" % b])
        assert I.digest(arrs) == str(g["digests"][b]), "batch %d differs from the reference" % b
    for m in range(len(al)):
        _state_equal(f.export(m), g, m)


def test_streaming_variants(oracle):
    kind, N, E, seed, bs, k, al, be, _ = I.STREAM_CASES["tiny_general"]
    g = golden("g1_variants")
    src, dst, neg, ts, eidx = I.make_stream(kind, N, E, seed)
    f = oracle.TpprOracle(N, k, len(al), al, be)
    on, oe, od, ow = f.streaming_topk_no_fake(np.concatenate([src[:40], dst[:40]]), ts[:40], eidx[:40])
    assert np.array_equal(np.stack(on), g["nf_nodes"]) and np.array_equal(np.stack(oe), g["nf_eidx"])
    assert np.array_equal(np.stack(od), g["nf_dt"]) and np.array_equal(np.stack(ow), g["nf_w"])
    sn, se, sd, sw = f.single_streaming_topk(np.concatenate([src[40:80], dst[40:80], neg[40:80]]), ts[40:80],
                                             eidx[40:80], 1)
    assert np.array_equal(sn, g["single_nodes"]) and np.array_equal(se, g["single_eidx"])
    assert np.array_equal(sd, g["single_dt"]) and np.array_equal(sw, g["single_w"])
    for m in range(len(al)):
        _state_equal(f.export(m), g, m)


def test_fill_matches_compute_val_tppr(oracle):
    kind, N, E, seed, bs, k, al, be, _ = I.STREAM_CASES["bip_k20"]
    g = golden("g2_fill")
    src, dst, neg, ts, eidx = I.make_stream(kind, N, E, seed)
    f = oracle.TpprOracle(N, k, len(al), al, be)
    f.update_only(src[:1000], dst[:1000], ts[:1000], eidx[:1000])
    for m in range(len(al)):
        _state_equal(f.export(m), g, m)
    # the reference's backup/restore alias the live state
    assert bool(g["alias"]) and bool(g["restore_is_noop"])


def test_hub_case_exercises_unstable_ties(oracle):
    """The hub fixture must contain sorts where numba's quicksort and a stable
    sort disagree, otherwise it does not pin the tie semantics."""
    rng = np.random.RandomState(0)
    diff = 0
    for _ in range(200):
        v = rng.randint(0, 4, 41).astype(np.float64)
        if not np.array_equal(oracle.numba_argsort(v), np.argsort(v, kind="stable")):
            diff += 1
    assert diff > 0


@pytest.mark.parametrize("name", list(I.PRUNE_CASES))
def test_pruning_matches_reference(oracle, name):
    kind, N, E, seed, nq, width, depth, k, alpha, beta = I.PRUNE_CASES[name]
    g = golden("g3_prune_" + name)
    src, dst, neg, ts, eidx = I.make_stream(kind, N, E, seed)
    csr = oracle.CsrOracle(src, dst, eidx, ts)
    for v in g["probe"]:
        nb, ei, tt = csr.find_before(int(v), np.inf)
        assert np.array_equal(nb, g["adj%d_nbr" % v]) and np.array_equal(ei, g["adj%d_eid" % v])
        assert np.array_equal(tt, g["adj%d_ts" % v])
    on = np.zeros((nq, k), np.int32)
    oe = np.zeros((nq, k), np.int32)
    od = np.zeros((nq, k), np.float32)
    ow = np.zeros((nq, k), np.float32)
    csr.get_pruned_topk(g["q_nodes"], g["q_ts"], width, depth, alpha, beta, k, on, oe, od, ow)
    assert np.array_equal(on, g["nodes"]) and np.array_equal(oe, g["eidx"])
    assert np.array_equal(od, g["dt"]) and np.array_equal(ow, g["w"])


def test_numba_int_pow_matches_python_restatement(oracle):
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import numba_semantics as ns
    for beta in (0.5, 0.95, 0.9, 0.7, 0.999):
        for n in (0, 1, 2, 3, 7, 20, 63, 64, 1000, 65536, 65537, 200000):
            assert oracle.numba_int_pow(beta, n) == ns.numba_int_pow(beta, n)
    rng = np.random.RandomState(3)
    for n in (1, 2, 15, 16, 17, 41, 81, 155, 420):
        for _ in range(20):
            v = np.round(rng.random_sample(n), 1 if _ % 2 else 6)
            assert np.array_equal(oracle.numba_argsort(v), ns.numba_argsort(v))


def test_argsort_restatements_match_numba_source(oracle):
    """The tie order of every fixture (and of the C oracle, and through it of the HIP kernels) rests on the
    restatement of numba's argsort quicksort.  numba's own numba/misc/quicksort.py (0.54.1, the copy on disk in the
    build container) is pure Python: oracle/numba_pin.py loads it standalone and runs it beside both restatements on
    5 240 tie-heavy arrays (n = 1..129, 155, 420).  Skipped where no numba tree exists (the GPU box)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import numba_pin
    if numba_pin.quicksort_path() is None:
        pytest.skip("no numba/misc/quicksort.py on this machine")
    n, bad = numba_pin.compare(extra=[oracle.numba_argsort])
    assert n >= 5000 and bad == 0, "%d of %d arrays sorted differently from numba's quicksort.py" % (bad, n)


@pytest.mark.parametrize("name", list(I.EMBED_CASES))
def test_embedding_and_memory_protocol(oracle, name):
    N, E, D, F, T, k, al, be, seed, bs, nb = I.EMBED_CASES[name]
    g = golden("g45_embed_" + name)
    src, dst, neg, ts, eidx = I.make_stream("general", N, E, seed)
    w = I.model_weights(D, F, T, len(al), seed)
    mem0, efeat = I.random_tables(N, E + 1, D, F, seed)
    tw = I.time_encode_weights(T)
    M = len(al)

    # G4: embedding from a random memory table
    f = oracle.TpprOracle(N, k, M, al, be)
    half = E // 2
    f.update_only(src[:half], dst[:half], ts[:half], eidx[:half])
    s, e = half, half + bs
    nodes = np.concatenate([src[s:e], dst[s:e], neg[s:e]])
    on, oe, od, ow = f.streaming_topk(nodes, ts[s:e], eidx[s:e])
    emb = oracle.embed(mem0, efeat, tw, nodes, np.stack(on), np.stack(oe), np.stack(od), np.stack(ow), w)
    assert np.abs(emb - g["emb"]).max() <= EMB_TOL
    avg = np.mean(np.sum(ow[0][:2 * bs], axis=1))
    assert abs(avg - float(g["average_topk"])) < 1e-6

    # G5: eval-mode protocol (model/tgn_model.py:124-174), 4 consecutive batches
    f = oracle.TpprOracle(N, k, M, al, be)
    mem = oracle.MemoryOracle(N, D, 2 * D + F + T)
    gru = {kk: w[kk] for kk in ("w_ih", "w_hh", "b_ih", "b_hh")}
    aff = dict(fc1_w=w["aff1_w"], fc1_b=w["aff1_b"], fc2_w=w["aff2_w"], fc2_b=w["aff2_b"])
    mem.gru_update(gru, None)                        # first eval batch flushes pending messages
    for b in range(nb):
        s, e = b * bs, (b + 1) * bs
        nodes = np.concatenate([src[s:e], dst[s:e], neg[s:e]])
        on, oe, od, ow = f.streaming_topk(nodes, ts[s:e], eidx[s:e])
        emb = oracle.embed(mem.memory, efeat, tw, nodes, np.stack(on), np.stack(oe), np.stack(od), np.stack(ow), w)
        assert np.abs(emb - g["eval_b%d_emb" % b]).max() <= EMB_TOL, "batch %d" % b
        B = e - s
        prob = oracle.affinity(np.concatenate([emb[:B], emb[:B]]), np.concatenate([emb[B:2 * B], emb[2 * B:]]), aff)
        assert np.abs(prob - g["eval_b%d_prob" % b]).max() <= EMB_TOL
        mem.store_messages(efeat, tw, src[s:e], dst[s:e], ts[s:e], eidx[s:e])
        mem.gru_update(gru, np.unique(np.concatenate([src[s:e], dst[s:e]])))
    pre = "eval_b%d_" % (nb - 1)
    assert np.abs(mem.memory - g[pre + "memory"]).max() <= EMB_TOL
    assert np.array_equal(mem.last_update, g[pre + "last_update"])
    assert np.abs(mem.messages - g[pre + "messages"]).max() <= EMB_TOL
    assert np.array_equal(mem.timestamps, g[pre + "timestamps"])
    assert np.array_equal(mem.flags, g[pre + "flags"])


def test_time_encode(oracle):
    g = golden("g6_timeencode")
    assert np.array_equal(I.time_encode_weights(100), g["time_w"])
    enc = np.cos(g["dts"][..., None] * g["time_w"][None, None, :]).astype(np.float32)
    assert np.abs(enc - g["enc"]).max() <= 1e-6


@pytest.mark.parametrize("name", list(I.ATTENTION_CASES))
def test_attention_restatement_matches_reference_layer(name):
    """oracle/attention_np.py vs the outputs of the reference's own TemporalAttentionLayer (g9)."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import attention_np
    D, F, T, k, N, heads, seed = I.ATTENTION_CASES[name]
    g = golden("g9_attention_" + name)
    out, aw = attention_np.temporal_attention(*I.attention_inputs(D, F, T, k, N, seed), I.attention_weights(D, F, T, seed),
                                              heads)
    assert np.abs(out - g["out"]).max() <= 1e-5
    assert np.abs(aw - g["attn_w"]).max() <= 1e-6
    assert np.abs(g["attn_w"][3]).max() == 0 and np.abs(aw[3]).max() == 0       # the row without neighbours


@pytest.mark.parametrize("mode", ["eval", "train"])
@pytest.mark.parametrize("name", list(I.PRUNE_EMBED_CASES))
def test_pruning_protocol(oracle, name, mode):
    """tppr_strategy='pruning' through the whole per-batch protocol (config C4's shape), with the
    neighbour-finder swap of train.py:191,245 half-way."""
    kind, N, E, D, F, T, k, al, be, width, depth, seed, bs, nb, first, n_train = I.PRUNE_EMBED_CASES[name]
    g = golden("g45_prune_" + name)
    src, dst, neg, ts, eidx = I.make_stream(kind, N, E, seed)
    w = I.model_weights(D, F, T, len(al), seed)
    mem0, efeat = I.random_tables(N, E + 1, D, F, seed)
    nf_full = oracle.CsrOracle(src, dst, eidx, ts, N)
    nf_part = oracle.CsrOracle(src[:n_train], dst[:n_train], eidx[:n_train], ts[:n_train], N)
    p = oracle.ProtocolOracle(N, D, F, T, k, al, be, w, efeat, I.time_encode_weights(T), "pruning", nf_part, width, depth)
    p.mem.memory[...] = mem0
    train = mode == "train"
    for b in range(nb):
        if b == nb // 2:
            p.set_neighbor_finder(nf_full)
        s, e = first + b * bs, first + (b + 1) * bs
        emb, prob = p.batch(src[s:e], dst[s:e], neg[s:e], ts[s:e], eidx[s:e], train)
        assert np.abs(emb - g["%s_b%d_emb" % (mode, b)]).max() <= EMB_TOL, "batch %d" % b
        assert np.abs(prob - g["%s_b%d_prob" % (mode, b)]).max() <= EMB_TOL
    assert abs(p.average_topk - float(g["%s_average_topk" % mode])) < 1e-6
    pre = "%s_b%d_" % (mode, nb - 1)
    assert np.abs(p.mem.memory - g[pre + "memory"]).max() <= EMB_TOL
    assert np.array_equal(p.mem.last_update, g[pre + "last_update"])
    assert np.abs(p.mem.messages - g[pre + "messages"]).max() <= EMB_TOL
    assert np.array_equal(p.mem.timestamps, g[pre + "timestamps"])
    assert np.array_equal(p.mem.flags, g[pre + "flags"])


class _OracleEpochAdapter:
    """The oracle under the epoch protocol, with the reference's aliasing: backups of the
    T-PPR state are the live object, restore_val_tppr makes the val state live."""

    def __init__(self, oracle, case, w, efeat):
        N, E, D, F, T, k, al, be = case[:8]
        self.o = oracle
        self.p = oracle.ProtocolOracle(N, D, F, T, k, al, be, w, efeat, I.time_encode_weights(T))
        self.val = None
        self.M = len(al)

    def init_memory(self):
        self.p.init_memory()

    def reset_tppr(self):                           # new objects; the old ones live on in val / backups
        p = self.p
        p.tppr = self.o.TpprOracle(p.N, p.k, self.M, p.alpha, p.beta)

    def fill_tppr(self, src, dst, ts, eidx, filled):
        if filled:
            self.p.tppr = self.val                  # restore_val_tppr: shallow copy == alias
        else:
            self.p.tppr.update_only(src, dst, ts, eidx)
            self.val = self.p.tppr

    def backup_tppr(self):
        return self.p.tppr

    def restore_tppr(self, b):
        self.p.tppr = b

    def backup_memory(self):
        return self.p.backup_memory()

    def restore_memory(self, b):
        self.p.restore_memory(b)

    def batch(self, src, dst, neg, ts, eidx, train):
        return self.p.batch(src, dst, neg, ts, eidx, train)[1]

    def memory_state(self):
        m = self.p.mem
        return dict(memory=m.memory, last_update=m.last_update, messages=m.messages, timestamps=m.timestamps,
                    flags=m.flags)

    def tppr_state(self):
        return {"m%d_%s" % (m, kk): v for m in range(self.M) for kk, v in self.p.tppr.export(m).items()}


@pytest.mark.parametrize("name", list(I.EPOCH_CASES))
def test_epoch_protocol(oracle, name):
    """Two epochs of train -> fill_tppr -> val -> backup/restore -> inductive val, then test
    (train.py:188-191,241-269,296-306) against the reference's own run (g10)."""
    case = I.EPOCH_CASES[name]
    N, E, D, F, T, k, al, be, seed = case[:9]
    g = golden("g10_epoch_" + name)
    streams = I.make_stream("general", N, E, seed)
    w = I.model_weights(D, F, T, len(al), seed)
    _, efeat = I.random_tables(N, E + 1, D, F, seed)
    epoch_protocol(_OracleEpochAdapter(oracle, case, w, efeat), g, case, streams, make_checker(g, EMB_TOL))


def test_torch_cpu_p23_equals_c_port(oracle):
    """oracle/torch_cpu.py (torch-CPU ops: bench.py's cpu_baseline for P2 / P3) against the C port on the same
    stream: embeddings, memory, messages within 1e-5; last_update and flags exact."""
    N, E, D, F, T, k, al, be, seed, bs, nb = I.EMBED_CASES["d100_f172"]
    src, dst, neg, ts, eidx = I.make_stream("general", N, E, seed)
    w = I.model_weights(D, F, T, len(al), seed)
    _, efeat = I.random_tables(N, E + 1, D, F, seed)
    tw = I.time_encode_weights(T)
    a = oracle.ProtocolOracle(N, D, F, T, k, al, be, w, efeat, tw)
    b = oracle.ProtocolOracle(N, D, F, T, k, al, be, w, efeat, tw, n_threads=2)
    b.p23 = "torch"
    for q in range(E // bs):
        s, e = q * bs, (q + 1) * bs
        ea, _ = a.batch(src[s:e], dst[s:e], neg[s:e], ts[s:e], eidx[s:e], False)
        eb, _ = b.batch(src[s:e], dst[s:e], neg[s:e], ts[s:e], eidx[s:e], False)
        assert np.abs(ea - eb).max() <= 1e-5, "batch %d" % q
    assert np.abs(a.mem.memory - b.mem.memory).max() <= 1e-5
    assert np.abs(a.mem.messages - b.mem.messages).max() <= 1e-5
    assert np.array_equal(a.mem.last_update, b.mem.last_update)
    assert np.array_equal(a.mem.flags, b.mem.flags)


def test_oracle_under_sanitizers():
    """SURVEY.md section 5 (race detection / sanitizers): the C oracle built with -fsanitize=address,undefined
    (`make -C oracle sanitize`) and driven through every entry point -- hubs, ties, self-loops, duplicate
    timestamps, empty rows, rejected out-of-range ids, OpenMP paths -- must finish without a report.  (GPU
    AddressSanitizer is not available on this pool; the host-side checker is what can be sanitized.)"""
    import os
    import subprocess
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
    subprocess.run(["make", "-C", root, "sanitize"], check=True, stdout=subprocess.DEVNULL)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([os.path.join(root, "_san", "sanitize_driver")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    assert "sanitize_driver: ok" in r.stdout and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr


def test_torch_cpu_training_step_matches_reference_gradients(oracle):
    """oracle/torch_cpu_train.py (the CPU baseline of bench.py's training leg) against fixture g8_train_grads: loss and every
    parameter gradient of four training steps of the reference itself (train.py:205-215, no optimizer step, dropout 0)."""
    import torch_cpu_train
    name = "d20_f7"
    N, E, D, F, T, k, al, be, seed, bs, nb = I.EMBED_CASES[name]
    g = golden("g8_train_grads")
    src, dst, neg, ts, eidx = I.make_stream("general", N, E, seed)
    w = I.model_weights(D, F, T, len(al), seed)
    _, efeat = I.random_tables(N, E + 1, D, F, seed)
    tr = torch_cpu_train.TorchCpuTrainer(N, D, F, T, k, al, be, w, efeat, I.time_encode_weights(T), dropout=0.0, n_threads=1,
                                         optimizer=False)
    seen = 0
    for b in range(nb):
        s, e = b * bs, (b + 1) * bs
        loss = tr.step(src[s:e], dst[s:e], neg[s:e], ts[s:e], eidx[s:e], optimize=False)
        assert abs(loss - float(g["b%d_loss" % b])) <= 1e-5, "loss of batch %d" % b
        for pn, p in tr.p.items():
            key = "b%d_grad_%s" % (b, pn)
            if key in g.files:
                assert p.grad is not None, pn
                want = g[key]
                err = np.abs(p.grad.detach().numpy() - want).max()
                assert err <= 1e-5 + 1e-4 * np.abs(want).max(), "%s in batch %d: %g" % (pn, b, err)
                seen += 1
            else:
                assert p.grad is None or float(p.grad.abs().max()) == 0.0, pn
    assert seen >= 12 * nb
