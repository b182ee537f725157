"""Link-prediction scoring without a per-batch host round trip.

The reference's ``eval_edge_prediction`` (evaluation/evaluation.py:7-48) copies the probabilities
of every batch to the host and calls scikit-learn three times per batch.  Here the batch stays on
the device: the metrics below are torch ops (float64, the arithmetic scikit-learn does) on whatever
device the scores live on, the per-batch values are accumulated there, and the host reads three
numbers at the end.  Definitions follow scikit-learn exactly, ties included:

  average_precision  sum over distinct thresholds of (R_n - R_{n-1}) * P_n
                     (sklearn.metrics.average_precision_score)
  roc_auc            trapezoid under the ROC curve over distinct thresholds
                     (sklearn.metrics.roc_auc_score; equals P(pos > neg) + P(pos == neg) / 2)
  accuracy           np.argmax(np.hstack([pos, neg]), axis=1) == 0, i.e. pos >= neg
                     (evaluation/evaluation.py:40-45)
"""
import math

import numpy as np
import torch


def _curve(pos, neg):
    """Scores descending: cumulative true / false positives, the mask of the last element of every run
    of equal scores (= the distinct thresholds) and, per element, the index before its run starts
    (-1 for the first run).  Everything is a fixed-shape device op: no host synchronisation."""
    score = torch.cat([pos.reshape(-1), neg.reshape(-1)]).to(torch.float64)
    label = torch.cat([torch.ones(pos.numel(), dtype=torch.float64, device=score.device),
                       torch.zeros(neg.numel(), dtype=torch.float64, device=score.device)])
    order = torch.argsort(score, descending=True, stable=True)
    score, label = score[order], label[order]
    tps = torch.cumsum(label, 0)
    fps = torch.cumsum(1.0 - label, 0)
    idx = torch.arange(score.numel(), device=score.device)
    first = torch.ones_like(score, dtype=torch.bool)
    first[1:] = score[1:] != score[:-1]
    last = torch.ones_like(score, dtype=torch.bool)
    last[:-1] = first[1:]
    before = torch.cummax(torch.where(first, idx, torch.zeros_like(idx)), 0).values - 1
    return tps, fps, last, before


def _at(x, before):
    """x[before], 0 where before == -1."""
    return torch.where(before >= 0, x[before.clamp(min=0)], torch.zeros_like(x))


def average_precision(pos, neg):
    tps, fps, last, before = _curve(pos, neg)
    recall = tps / float(pos.numel())
    precision = tps / (tps + fps)
    term = (recall - _at(recall, before)) * precision
    return torch.sum(torch.where(last, term, torch.zeros_like(term)))


def roc_auc(pos, neg):
    tps, fps, last, before = _curve(pos, neg)
    tpr, fpr = tps / float(pos.numel()), fps / float(neg.numel())
    term = (fpr - _at(fpr, before)) * (tpr + _at(tpr, before)) * 0.5
    return torch.sum(torch.where(last, term, torch.zeros_like(term)))


def accuracy(pos, neg):
    return (pos.reshape(-1) >= neg.reshape(-1)).to(torch.float64).mean()


def link_metrics(pos, neg, out=None):
    """(average_precision, roc_auc, accuracy) of one batch as a float64[3] tensor on the scores' device.  CUDA scores of
    up to 8192 pairs go through ONE HIP kernel (zt_link_metrics, csrc/scoring.hip: bitonic sort in LDS + the two curve
    sums; ``out`` given: the values are ADDED to it there, no extra op); anything else through the torch ops above --
    the same definitions, which the CPU tests hold against scikit-learn and the GPU tests against each other."""
    pos, neg = pos.reshape(-1), neg.reshape(-1)
    if pos.is_cuda and pos.dtype == torch.float32 and neg.dtype == torch.float32 and 0 < pos.numel() == neg.numel() <= 8192:
        import ctypes as C
        from ._capi import check, lib, ptr, stream_ptr
        acc = out if out is not None else torch.zeros(3, dtype=torch.float64, device=pos.device)
        check(lib().zt_link_metrics(ptr(pos.contiguous()), ptr(neg.contiguous()), C.c_int64(pos.numel()), ptr(acc),
                                    C.c_int32(1 if out is not None else 0), stream_ptr()), "zt_link_metrics")
        return acc
    m = torch.stack([average_precision(pos, neg), roc_auc(pos, neg), accuracy(pos, neg)])
    if out is not None:
        out += m
        return out
    return m


@torch.no_grad()
def eval_edge_prediction(model, negative_edge_sampler, data, n_neighbors, batch_size):
    """evaluation/evaluation.py:7-48 with the same protocol and return value (mean AP, mean AUC, mean
    accuracy over the batches); probabilities and metrics stay on the device, the host reads once."""
    assert negative_edge_sampler.seed is not None
    negative_edge_sampler.reset_random_state()
    model = model.eval()
    n = data.n_interactions
    nb = math.ceil(n / batch_size)
    acc = None
    for b in range(nb):
        s, e = b * batch_size, min(n, (b + 1) * batch_size)
        _, negatives = negative_edge_sampler.sample(e - s)
        pos, neg = model.compute_edge_probabilities(data.sources[s:e], data.destinations[s:e], negatives,
                                                    data.timestamps[s:e], data.edge_idxs[s:e], n_neighbors, train=False)
        if acc is None:
            acc = torch.zeros(3, dtype=torch.float64, device=pos.device)
        link_metrics(pos, neg, out=acc)
    out = (acc / nb).cpu().numpy()
    return float(out[0]), float(out[1]), float(out[2])
