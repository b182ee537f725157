"""zebra_amd.evaluation's metrics against scikit-learn (the functions the reference calls,
evaluation/evaluation.py:44-46), on CPU tensors, with heavy ties."""
import numpy as np
import pytest
import torch

sk = pytest.importorskip("sklearn.metrics")


@pytest.mark.parametrize("seed,levels", [(0, 0), (1, 7), (2, 2), (3, 50)])
def test_metrics_match_sklearn(seed, levels):
    from zebra_amd import evaluation as ev
    rng = np.random.RandomState(seed)
    n = 257
    pos, neg = rng.random_sample(n) * 0.7 + 0.3, rng.random_sample(n) * 0.8
    if levels:                                  # quantise: many equal scores across and within the classes
        pos, neg = np.round(pos * levels) / levels, np.round(neg * levels) / levels
    pos, neg = pos.astype(np.float32), neg.astype(np.float32)
    y = np.concatenate([np.ones(n), np.zeros(n)])
    sc = np.concatenate([pos, neg])
    tp, tn = torch.from_numpy(pos).reshape(-1, 1), torch.from_numpy(neg).reshape(-1, 1)
    assert abs(float(ev.average_precision(tp, tn)) - sk.average_precision_score(y, sc)) < 1e-12
    assert abs(float(ev.roc_auc(tp, tn)) - sk.roc_auc_score(y, sc)) < 1e-12
    want_acc = sk.accuracy_score(np.zeros(n), np.argmax(np.hstack([pos.reshape(-1, 1), neg.reshape(-1, 1)]), axis=1))
    assert abs(float(ev.accuracy(tp, tn)) - want_acc) < 1e-12
