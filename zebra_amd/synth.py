"""Synthetic temporal edge streams.

No dataset ships with the build and none can be downloaded, so tests, the
golden-vector generator and bench.py all draw streams from here.  Shapes follow
the reference's on-disk format (utils/preprocess_data.py:30-79): node ids and
edge ids are 1-based, id 0 is padding, bipartite graphs put users in
1..U and items in U+1..U+I; timestamps are non-decreasing float64 seconds.
"""
import numpy as np


def _power_law_sampler(n, s, rng, perm_seed=None):
    """Return draw(m) -> int64[m] ids in [0, n) whose popularity RANK r has p(r) ~ (r+1)^-s.
    perm_seed=None: id == rank (the hot ids are the low ones, contiguous in every per-node table);
    otherwise ids are a seeded random permutation of the ranks, so hot rows are scattered."""
    p = np.arange(1, n + 1, dtype=np.float64) ** (-s)
    cdf = np.cumsum(p)
    cdf /= cdf[-1]
    perm = None if perm_seed is None else np.random.RandomState(perm_seed).permutation(n)

    def draw(m):
        r = np.minimum(np.searchsorted(cdf, rng.random_sample(m)), n - 1)
        return r if perm is None else perm[r]
    return draw


def power_law_stream(n_nodes, n_edges, bipartite=None, s=0.9, seed=2020, mean_dt=30.0,
                     t0=0.0, first_eidx=1, perm_seed=None):
    """Bounded power-law temporal stream.

    n_nodes  : number of real nodes (ids 1..n_nodes); ignored when bipartite.
    bipartite: (U, I) -> sources in 1..U, destinations in U+1..U+I.
    Returns src int32[E], dst int32[E], ts float64[E] (strictly increasing),
    eidx int64[E] = first_eidx..first_eidx+E-1.
    perm_seed: shuffle which node id carries which popularity rank (bench.py does; None keeps id == rank,
    which the committed fixtures and tests were generated with).
    """
    rng = np.random.RandomState(seed)
    if bipartite is not None:
        U, I = bipartite
        src = 1 + _power_law_sampler(U, s, rng, perm_seed)(n_edges)
        dst = 1 + U + _power_law_sampler(I, s, rng, None if perm_seed is None else perm_seed + 1)(n_edges)
    else:
        draw = _power_law_sampler(n_nodes, s, rng, perm_seed)
        src = 1 + draw(n_edges)
        dst = 1 + draw(n_edges)
    ts = t0 + np.cumsum(rng.exponential(mean_dt, n_edges))
    eidx = np.arange(first_eidx, first_eidx + n_edges, dtype=np.int64)
    return src.astype(np.int32), dst.astype(np.int32), ts.astype(np.float64), eidx


def negatives(dst, n, seed=0):
    """RandEdgeSampler.sample's destination half (utils/util.py:69-84): uniform
    over the observed (unique) destinations."""
    rng = np.random.RandomState(seed)
    uniq = np.unique(dst)
    return uniq[rng.randint(0, len(uniq), n)].astype(np.int32)


def edge_features(n_edges_plus1, F, seed=1):
    """[|E|+1, F] float32; row 0 is the padding row (zeros).  F == 1 means the
    dataset has no edge features and the reference substitutes zeros
    (train.py:133-136)."""
    if F == 1:
        return np.zeros((n_edges_plus1, 1), np.float32)
    rng = np.random.RandomState(seed)
    f = rng.standard_normal((n_edges_plus1, F)).astype(np.float32)
    f[0] = 0
    return f


# Shapes of BASELINE.json's configs (SURVEY.md section 8, table at the top).
WORKLOADS = {
    # name: dict(nodes, edges, bipartite, F, bs, k, strategy, alpha, beta)
    "c1": dict(bipartite=(8227, 1000), n_nodes=9227, n_edges=157474, F=172, bs=200, k=20,
               strategy="streaming", alpha=[0.1], beta=[0.9]),
    "c2": dict(bipartite=(8227, 1000), n_nodes=9227, n_edges=157474, F=172, bs=200, k=20,
               strategy="streaming", alpha=[0.1, 0.1], beta=[0.5, 0.95]),
    "c3": dict(bipartite=(10000, 984), n_nodes=10984, n_edges=672447, F=172, bs=600, k=20,
               strategy="streaming", alpha=[0.1, 0.1], beta=[0.5, 0.95]),
    "c4": dict(bipartite=None, n_nodes=194085, n_edges=1443339, F=1, bs=1000, k=40,
               strategy="pruning", width=10, depth=2, alpha=[0.1, 0.1], beta=[0.5, 0.95]),
    "c5": dict(bipartite=None, n_nodes=10_000_000, n_edges=100_000_000, F=1, bs=4096, k=20,
               strategy="streaming", alpha=[0.1, 0.1], beta=[0.5, 0.95]),
}


def pipeline_settings(wl, steps, tppr_cus=-1, group=-1):
    """(tppr_cus, group) for TGN.enable_pipeline on workload ``wl`` over a timed region of ``steps`` batches:
    the ONE place the choice is made -- bench.py and the per-config parity tests (tests/test_configs_gpu.py)
    both call it, so the tests run exactly the launch configuration that is timed.
    group: as many batches per T-PPR launch as fit (<= 16384 edges, <= 8: a launch's start, tail and the gap to the
    next one -- ~110 us -- are paid once per launch; C3 0.149 -> 0.141 ms/step from 4 to 8, worse again at 16); with
    ZT_RELEASE_LAUNCH fewer for a short region (the first and the last batches of a region are then queried one by one, which a
    20-step run pays for with large groups);
    1 for the pruning strategy.  tppr_cus: whole XCDs (32 CUs, one L2 each: a mask that splits an XCD costs the
    aggregation 55-75 %) -- two for the T-PPR stream: 16 hub chains x 2 models + a general queue need 48 workgroups
    (on one XCD only 10 chains fit, measured 0.181 -> 0.153 ms/step on C3); at C5's batch the general queue needs ~53
    of the 64 workgroups itself and the prepass keeps 5 chains per model (tppr_prepass.hip: d_chain_budget), which
    costs k_stream 3 % against three XCDs with 16 chains and gives the main stream -- the bound -- 192 CUs
    (aggregation 214 -> 176 us): C5 0.341 -> 0.333 ms/step at 200 steps, 0.385 -> 0.379 at 20 (one box, two runs each,
    profiles/r4/experiments/cus_and_chains_c5.log); no masks for the pruning strategy, whose query kernel wants the
    whole chip."""
    streaming = wl["strategy"] == "streaming"
    if group < 1:
        from . import _capi
        if _capi.kernel_choice(_capi.CHOICE_GROUP_RELEASE) == _capi.RELEASE_LAUNCH:       # (RELEASE_LAUNCH_FULL: full groups, as by member)
            group = max(1, min(8, 16384 // wl["bs"], max(1, steps // 10))) if streaming else 1
        else:
            # round 6: the batches of a launch are released to the aggregation one by one (pipeline.hip, "release by member"),
            # so the first and last batches of a short region no longer pay for large groups: full groups whatever `steps` is
            group = max(1, min(8, 16384 // wl["bs"])) if streaming else 1
    if tppr_cus < 0:
        tppr_cus = 64 if streaming else 0
    return tppr_cus, group


def pipeline_look(group):
    """Batches a step should see ahead of the current one so that T-PPR launch groups stay full: the rest of this group, the
    next group, the one after it, and one more (a group takes at most n - 1 of the n followers in sight along:
    pipeline.hip, make_group)."""
    return 3 * group + 1
