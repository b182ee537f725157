"""Deterministic inputs shared by the golden-vector generator and the tests.

The fixtures under tests/golden/*.npz hold EXPECTED OUTPUTS (and small inputs);
bulky inputs (random tables, weights) are regenerated here from fixed seeds with
numpy's frozen legacy RandomState, so generator and tests see identical bytes.
Own code; nothing here comes from the reference.
"""
import hashlib

import numpy as np

# ----------------------------------------------------------------------------
# streams
# ----------------------------------------------------------------------------


def _pl(rng, n, m, s=0.9):
    p = np.arange(1, n + 1, dtype=np.float64) ** (-s)
    cdf = np.cumsum(p) / p.sum()
    return np.minimum(np.searchsorted(cdf, rng.random_sample(m)), n - 1)


def make_stream(kind, n_nodes, n_edges, seed):
    """Return src, dst, neg (int32), ts (float64), eidx (int64); node ids in
    1..n_nodes-1 (n_nodes counts the padding id 0, like args.n_nodes)."""
    rng = np.random.RandomState(seed)
    real = n_nodes - 1
    if kind == "general":
        src = 1 + _pl(rng, real, n_edges)
        dst = 1 + _pl(rng, real, n_edges)
        # self-loops, repeated pairs and a hub are all wanted
        loops = rng.random_sample(n_edges) < 0.03
        dst[loops] = src[loops]
        rep = np.where(rng.random_sample(n_edges) < 0.05)[0]
        rep = rep[rep > 0]
        src[rep] = src[rep - 1]
        dst[rep] = dst[rep - 1]
        ts = np.cumsum(rng.exponential(30.0, n_edges))
        # duplicate timestamps (across different nodes and within a node)
        dup = np.where(rng.random_sample(n_edges) < 0.1)[0]
        dup = dup[dup > 0]
        ts[dup] = ts[dup - 1]
        ts = np.maximum.accumulate(ts)
    elif kind == "bipartite":
        U = (real * 5) // 6
        I = real - U
        src = 1 + _pl(rng, U, n_edges)
        dst = 1 + U + _pl(rng, I, n_edges)
        ts = np.cumsum(rng.exponential(30.0, n_edges))
    elif kind == "hub":
        # one dominant hub so that candidate lists fill up and tie quickly
        src = 1 + _pl(rng, real, n_edges, s=1.6)
        dst = 1 + _pl(rng, real, n_edges, s=0.5)
        ts = np.cumsum(rng.exponential(30.0, n_edges))
    else:
        raise ValueError(kind)
    uniq = np.unique(dst)
    neg = uniq[rng.randint(0, len(uniq), n_edges)]
    eidx = np.arange(1, n_edges + 1, dtype=np.int64)
    return (src.astype(np.int32), dst.astype(np.int32), neg.astype(np.int32),
            ts.astype(np.float64), eidx)


# name -> (kind, n_nodes, n_edges, seed, bs, k, alpha_list, beta_list, store_full)
STREAM_CASES = {
    "tiny_general": ("general", 50, 600, 11, 7, 5, [0.1, 0.1], [0.5, 0.95], True),
    "bip_k20": ("bipartite", 241, 2000, 12, 200, 20, [0.1, 0.1], [0.5, 0.95], False),
    "gen_k40": ("general", 200, 2000, 13, 200, 40, [0.0, 0.3], [0.9, 0.7], False),
    "bs1_single": ("general", 50, 300, 14, 1, 5, [0.1], [0.9], True),
    "hub_ties": ("hub", 120, 1500, 15, 50, 20, [0.1], [0.5], False),
}

# name -> (stream kind, n_nodes, n_edges, seed, n_queries, width, depth, k, alpha, beta)
PRUNE_CASES = {
    "w10d2_k20": ("general", 300, 2000, 21, 240, 10, 2, 20, 0.1, 0.5),
    "w20d2_k20": ("general", 300, 2000, 22, 120, 20, 2, 20, 0.1, 0.95),
    "w5d3_k5": ("bipartite", 241, 1500, 23, 240, 5, 3, 5, 0.0, 0.9),
    "w10d2_k40": ("hub", 120, 1500, 24, 150, 10, 2, 40, 0.3, 0.7),
}


def prune_queries(src, dst, ts, n_nodes, nq, seed):
    """Query (node, time) pairs: batch-like rows, plus nodes with no history."""
    rng = np.random.RandomState(seed + 1000)
    E = len(src)
    pick = rng.randint(E // 4, E, nq)
    nodes = np.where(rng.random_sample(nq) < 0.5, src[pick], dst[pick]).astype(np.int32)
    qts = ts[pick].copy()
    # a few queries before any edge of that node, and one never-seen node id
    qts[: nq // 20] = ts[0] - 1.0
    nodes[nq // 20] = n_nodes - 1 if (n_nodes - 1) not in set(src.tolist()) | set(dst.tolist()) else nodes[nq // 20]
    return nodes, qts.astype(np.float64)


# ----------------------------------------------------------------------------
# model inputs
# ----------------------------------------------------------------------------


def model_weights(D, F, T, M, seed):
    """Random parameters in torch layout for the embedding module, the GRU
    updater and the link scorer."""
    rng = np.random.RandomState(seed)

    def lin(o, i):
        std = np.sqrt(2.0 / (o + i))
        return (rng.standard_normal((o, i)) * std).astype(np.float32), \
               (rng.uniform(-1, 1, o) / np.sqrt(i)).astype(np.float32)

    w = {}
    w["fc1_w"], w["fc1_b"] = lin(D, D + F + T)
    w["fc2_w"], w["fc2_b"] = lin(D, D)
    w["fc1s_w"], w["fc1s_b"] = lin(D, D)
    w["fc2s_w"], w["fc2s_b"] = lin(D, D)
    msg = 2 * D + F + T
    s = 1.0 / np.sqrt(D)
    w["w_ih"] = rng.uniform(-s, s, (3 * D, msg)).astype(np.float32)
    w["w_hh"] = rng.uniform(-s, s, (3 * D, D)).astype(np.float32)
    w["b_ih"] = rng.uniform(-s, s, 3 * D).astype(np.float32)
    w["b_hh"] = rng.uniform(-s, s, 3 * D).astype(np.float32)
    H = D * (M + 1)
    w["aff1_w"], w["aff1_b"] = lin(H, 2 * H)
    w["aff2_w"], w["aff2_b"] = lin(1, H)
    return w


def time_encode_weights(T):
    """TimeEncode's frozen frequencies, evaluated exactly as the reference does
    (model/time_encoding.py:18): numpy, entirely in float32."""
    return (1 / 10 ** np.linspace(0, 9, T, dtype=np.float32)).astype(np.float32)


def random_tables(n_nodes, n_edges_plus1, D, F, seed):
    rng = np.random.RandomState(seed)
    mem = (rng.standard_normal((n_nodes, D)) * 0.5).astype(np.float32)
    mem[0] = 0
    if F == 1:
        ef = np.zeros((n_edges_plus1, 1), np.float32)
    else:
        ef = rng.standard_normal((n_edges_plus1, F)).astype(np.float32)
        ef[0] = 0
    return mem, ef


# name -> (n_nodes, n_edges, D, F, T, k, alpha_list, beta_list, seed, bs, n_batches)
EMBED_CASES = {
    "d100_f172": (120, 400, 100, 172, 100, 20, [0.1, 0.1], [0.5, 0.95], 31, 20, 4),
    "d100_f1": (120, 400, 100, 1, 100, 20, [0.1, 0.1], [0.5, 0.95], 32, 20, 4),
    "d20_f7": (60, 300, 20, 7, 20, 5, [0.2], [0.8], 33, 16, 4),
}


def digest(arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


# ---------------------------------------------------------------------------
# ingest (g7): a small bipartite table in the reference's ml_{name}.csv format
# ---------------------------------------------------------------------------
def make_ml_table(n_users=60, n_items=40, n_edges=1500, seed=4242):
    """u (1..U), i (U+1..U+I), ts (sorted, with repeats), label, idx (1..E) like utils/preprocess_data.py."""
    rng = np.random.RandomState(seed)
    u = 1 + np.minimum((rng.pareto(1.2, n_edges) * 3).astype(np.int64), n_users - 1)
    i = 1 + n_users + np.minimum((rng.pareto(1.5, n_edges) * 2).astype(np.int64), n_items - 1)
    ts = np.sort(np.round(rng.exponential(50.0, n_edges).cumsum(), 0)).astype(np.float64)
    label = (rng.random_sample(n_edges) < 0.03).astype(np.float64)
    idx = np.arange(1, n_edges + 1, dtype=np.int64)
    return u, i, ts, label, idx


def write_ml_csv(path, u, i, ts, label, idx):
    """Same layout as the reference's preprocessing output (unnamed index column first)."""
    with open(path, "w") as f:
        f.write(",u,i,ts,label,idx\n")
        for r in range(len(u)):
            f.write("%d,%d,%d,%r,%r,%d\n" % (r, u[r], i[r], float(ts[r]), float(label[r]), idx[r]))


# ---------------------------------------------------------------------------
# attention (g9): seeded weights + inputs of model/temporal_attention.py's layer
# ---------------------------------------------------------------------------
# name -> (D, F, T, k, N, heads, seed)
ATTENTION_CASES = {
    "d100_f172_k20": (100, 172, 100, 20, 96, 2, 51),
    "d100_f1_k10": (100, 1, 100, 10, 65, 2, 52),
    "d20_f4_k5_h4": (20, 4, 12, 5, 70, 4, 53),
}


def attention_weights(D, F, T, seed):
    """Parameters of TemporalAttentionLayer(n_node=D, n_nbr=D, n_edge=F, time=T, out=D) in torch layout."""
    rng = np.random.RandomState(seed)
    E, K = D + T, D + T + F

    def mat(o, i, s=None):
        s = np.sqrt(2.0 / (o + i)) if s is None else s
        return (rng.standard_normal((o, i)) * s).astype(np.float32)

    def vec(n, s=0.1):
        return (rng.uniform(-s, s, n)).astype(np.float32)

    return dict(q_w=mat(E, E), k_w=mat(E, K), v_w=mat(E, K), in_b=vec(3 * E), out_w=mat(E, E), out_b=vec(E),
                m1_w=mat(D, E + D), m1_b=vec(D), m2_w=mat(D, D), m2_b=vec(D))


def attention_inputs(D, F, T, k, N, seed):
    """src [N,D], src_time [N,1,T], nbr [N,k,D], nbr_time [N,k,T], edge [N,k,F], mask bool [N,k]
    (True = no neighbour); row 3 has no neighbour at all, row 5 all of them."""
    rng = np.random.RandomState(seed + 500)
    src = rng.standard_normal((N, D)).astype(np.float32)
    src_t = np.ones((N, 1, T), np.float32)                                    # cos(0)
    nbr = rng.standard_normal((N, k, D)).astype(np.float32)
    nbr_t = np.cos(rng.standard_normal((N, k, T)) * 5).astype(np.float32)
    edge = rng.standard_normal((N, k, F)).astype(np.float32)
    mask = rng.random_sample((N, k)) < 0.4
    mask[3] = True
    mask[5] = False
    mask[7, 1:] = True                                                        # exactly one neighbour
    return src, src_t, nbr, nbr_t, edge, mask


# ---------------------------------------------------------------------------
# pruning-strategy protocol (g45_prune_*): config C4's shape in small
# name -> (stream kind, n_nodes, n_edges, D, F, T, k, alpha, beta, width, depth, seed, bs, n_batches, first_edge,
#          train_finder_edges)
# ---------------------------------------------------------------------------
PRUNE_EMBED_CASES = {
    "d100_f1_k40": ("general", 150, 1200, 100, 1, 100, 40, [0.1, 0.1], [0.5, 0.95], 10, 2, 61, 16, 4, 900, 800),
    "d20_f7_k10": ("hub", 80, 800, 20, 7, 20, 10, [0.2], [0.8], 5, 3, 62, 12, 4, 500, 450),
}


# ---------------------------------------------------------------------------
# epoch protocol (g10): train.py:188-191,241-269,296-306 in small
# name -> (n_nodes, n_edges, D, F, T, k, alpha, beta, seed, bs, n_train, n_val, n_nn_val)
# ---------------------------------------------------------------------------
EPOCH_CASES = {
    "stream_d20_f7": (60, 520, 20, 7, 20, 5, [0.2, 0.1], [0.8, 0.5], 71, 16, 320, 96, 48),
}
