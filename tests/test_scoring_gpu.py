"""The link scorer and the link-prediction metrics on the device (csrc/scoring.hip)
against the oracle's MergeLayer (oracle/zo_nn.c: zo_affinity, pinned by the protocol fixtures), torch's own composition of
the reference's layers (model/tgn_model.py:185-188, utils/util.py:14-26) and scikit-learn's metric definitions
(evaluation/evaluation.py:34-45)."""
import numpy as np
import pytest
import torch

import inputs as I
from helpers import build_tgn

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _tgn(M, seed=5):
    D = T = 100
    al, be = ([0.1], [0.9]) if M == 1 else ([0.1, 0.1], [0.5, 0.95])
    w = I.model_weights(D, 1, T, M, seed)
    _, efeat = I.random_tables(50, 60, D, 1, seed)
    return build_tgn(50, 60, D, 1, T, 20, al, be, w, efeat).eval(), w


@pytest.mark.parametrize("M", [1, 2])
@pytest.mark.parametrize("B", [1, 15, 16, 17, 200, 511, 512, 1000, 2081, 4096, 8192, 8200])
def test_affinity_kernel_matches_oracle_and_torch(oracle, M, B):
    """zt_affinity -- the latency-organised kernel below 512 edges, the tiled one (16 edges per workgroup up to 8192, 32 beyond) from there -- for H = 200 (one T-PPR
    model: config C1) and H = 300 (two) over ragged and whole tiles: probabilities
    <= 1e-4 from the oracle's scorer and from torch's MergeLayer; two launches give the same bits (the N-tiles' partial
    scores are added in a fixed order whichever wave finishes last)."""
    tgn, w = _tgn(M)
    H = 100 * (M + 1)
    g = torch.Generator().manual_seed(B + M)
    emb = (torch.randn((3 * B, H), generator=g) * 0.7).cuda()
    got = tgn.score_device(emb)
    again = tgn.score_device(emb)
    torch.cuda.synchronize()
    assert torch.equal(got, again)
    with torch.no_grad():
        ref = tgn.affinity_score(torch.cat([emb[:B], emb[:B]]), emb[B:]).squeeze(1).sigmoid()
    assert float((got - ref).abs().max()) <= TOL
    e = emb.cpu().numpy()
    aff = dict(fc1_w=w["aff1_w"], fc1_b=w["aff1_b"], fc2_w=w["aff2_w"], fc2_b=w["aff2_b"])
    want = oracle.affinity(np.concatenate([e[:B], e[:B]]), e[B:], aff)         # (probabilities: zo_affinity ends with the sigmoid)
    assert np.abs(got.cpu().numpy() - want).max() <= TOL


def test_affinity_follows_weight_changes():
    """The packed copy of the scorer's weights is remade when a weight changes in place (optimizer step, load_state_dict)."""
    tgn, _ = _tgn(2)
    emb = torch.randn((3 * 64, 300), generator=torch.Generator().manual_seed(1)).cuda()
    a = tgn.score_device(emb)
    with torch.no_grad():
        tgn.affinity_score.fc1.weight.mul_(0.5)
        tgn.affinity_score.fc2.bias.add_(0.25)
        ref = tgn.affinity_score(torch.cat([emb[:64], emb[:64]]), emb[64:]).squeeze(1).sigmoid()
    b = tgn.score_device(emb)
    assert float((b - ref).abs().max()) <= TOL and float((a - b).abs().max()) > 1e-3


@pytest.mark.parametrize("B", [1, 2, 7, 150, 1000, 4096, 8192])
@pytest.mark.parametrize("ties", [False, True])
def test_link_metrics_kernel_matches_sklearn(B, ties):
    """zt_link_metrics (one kernel: sort, scans, curve sums in float64) against scikit-learn's average_precision_score /
    roc_auc_score / the reference's accuracy (evaluation/evaluation.py:40-45) and against the torch composition of
    zebra_amd.evaluation, with heavy ties (scores rounded to one decimal) and without; then the accumulating form."""
    sk = pytest.importorskip("sklearn.metrics")
    from zebra_amd import evaluation as ev
    rng = np.random.RandomState(B + (7 if ties else 0))
    pos = np.clip(rng.normal(0.65, 0.2, B), 0, 1).astype(np.float32)
    neg = np.clip(rng.normal(0.4, 0.2, B), 0, 1).astype(np.float32)
    if ties:
        pos, neg = np.round(pos, 1), np.round(neg, 1)
    p, n = torch.from_numpy(pos).cuda(), torch.from_numpy(neg).cuda()
    got = ev.link_metrics(p, n).cpu().numpy()
    y = np.concatenate([np.ones(B), np.zeros(B)])
    sc = np.concatenate([pos, neg]).astype(np.float64)
    want = [sk.average_precision_score(y, sc), sk.roc_auc_score(y, sc) if B > 0 else 0.0, float(np.mean(pos >= neg))]
    assert np.allclose(got, want, rtol=0, atol=1e-12), (got, want)
    comp = torch.stack([ev.average_precision(p, n), ev.roc_auc(p, n), ev.accuracy(p, n)]).cpu().numpy()
    assert np.allclose(got, comp, rtol=0, atol=1e-12)
    acc = torch.zeros(3, dtype=torch.float64, device="cuda")
    ev.link_metrics(p, n, out=acc)
    ev.link_metrics(p, n, out=acc)
    assert np.allclose(acc.cpu().numpy(), 2 * np.asarray(want), rtol=0, atol=1e-11)


@pytest.mark.parametrize("strategy", ["streaming"])
def test_pipeline_scores_every_whole_batch(strategy):
    """zt_pipeline_set_scoring: with TGN.enable_scoring the native step writes the batch's 2B probabilities behind its
    aggregation; they equal score_device on the embeddings the step returned, batch by batch, and follow a weight change."""
    name = "d100_f1"
    N, E, D, F, T, k, al, be, seed, bs, nb = I.EMBED_CASES[name]
    src, dst, neg, ts, eidx = I.make_stream("general", N, E, seed)
    w = I.model_weights(D, F, T, len(al), seed)
    _, efeat = I.random_tables(N, E + 1, D, F, seed)
    tgn = build_tgn(N, E + 1, D, F, T, k, al, be, w, efeat).eval()
    tgn.enable_pipeline(tppr_cus=0, max_batch=256, group=2)
    tgn.enable_scoring()
    dev = tgn.device
    t = [torch.from_numpy(x).to(dev) for x in (src, dst, neg, ts, eidx)]
    batches = [tuple(x[b * bs:(b + 1) * bs] for x in t) for b in range(E // bs)]
    with torch.cuda.stream(tgn.main_stream):
        for q, cur in enumerate(batches):
            if q == 2:
                with torch.no_grad():
                    tgn.affinity_score.fc1.bias.add_(0.1)
            emb = tgn.step_device(*cur, ahead=batches[q + 1: q + 4])
            got = tgn.last_prob().clone()
            want = tgn.score_device(emb)
            assert torch.equal(got, want), "batch %d" % q
    torch.cuda.synchronize()
    tgn.enable_pipeline(False)
