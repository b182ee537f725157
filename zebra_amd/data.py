"""Ingest of the reference's on-disk format.

``ml_{name}.csv`` (columns u, i, ts, label, idx; node ids 1-based, destinations offset by the number of
sources in bipartite datasets) and ``ml_{name}.npy`` (edge features, row 0 zeros) as written by
utils/preprocess_data.py:30-79.  ``get_data`` returns what utils/data_processing.py:80-149 returns --
the same six ``Data`` splits, the same inductive node sample (``random.seed(2020)`` +
``random.sample`` over the set of late nodes: reproduced with the same container operations, so the
sample is identical) -- but every per-edge Python loop of the reference (``Series.map(lambda ...)``,
the list comprehension over ``zip(sources, destinations)``) is a vectorised membership test, and
``compute_time_statistics`` is a grouped difference instead of two dictionaries.  The adjacency the
reference builds with Python lists (utils/util.py:90-107) is ``zebra_amd.tppr.get_neighbor_finder``.
"""
import os
import random

import numpy as np


class Data:
    """utils/data_processing.py:8-32 (same attributes)."""

    def __init__(self, sources, destinations, timestamps, edge_idxs, labels):
        self.sources = sources
        self.destinations = destinations
        self.timestamps = timestamps
        self.edge_idxs = edge_idxs
        self.labels = labels
        self.n_interactions = len(sources)
        self.unique_nodes = set(sources) | set(destinations)
        self.n_unique_nodes = len(self.unique_nodes)
        self.tbatch = None
        self.n_batch = 0

    def sample(self, ratio):
        size = int(ratio * self.n_interactions)
        inds = np.sort(random.sample(range(self.n_interactions), size))
        return Data(self.sources[inds], self.destinations[inds], self.timestamps[inds], self.edge_idxs[inds],
                    self.labels[inds])


def read_edge_csv(path):
    """The five columns of ml_{name}.csv as numpy arrays with pandas' dtypes (int64 ids, float64 ts)."""
    import pandas as pd
    df = pd.read_csv(path)
    return (df.u.values, df.i.values, df.ts.values, df.label.values, df.idx.values)


def load_feat(name, root="../data"):
    """utils/data_processing.py:70-79: (node features or None, edge features or None)."""
    node_p = os.path.join(root, name, "ml_%s_node.npy" % name)
    edge_p = os.path.join(root, name, "ml_%s.npy" % name)
    return (np.load(node_p) if os.path.exists(node_p) else None, np.load(edge_p) if os.path.exists(edge_p) else None)


def get_data(dataset_name, root="../data", verbose=False):
    """utils/data_processing.py:83-149; ``root`` replaces the hard-wired '../data'."""
    sources, destinations, timestamps, labels, edge_idxs = read_edge_csv(
        os.path.join(root, dataset_name, "ml_%s.csv" % dataset_name))
    val_time, test_time = list(np.quantile(timestamps, [0.70, 0.85]))
    full_data = Data(sources, destinations, timestamps, edge_idxs, labels)

    random.seed(2020)                                    # "ensure we get the same graph"
    node_set = set(sources) | set(destinations)
    n_total_unique_nodes = len(node_set)
    n_edges = len(sources)
    late = timestamps > val_time
    test_node_set = set(sources[late]).union(set(destinations[late]))
    # the reference samples from the set itself (Python < 3.11: its iteration order)
    new_test_node_set = set(random.sample(tuple(test_node_set), int(0.1 * n_total_unique_nodes)))
    new_test = np.fromiter(new_test_node_set, dtype=sources.dtype, count=len(new_test_node_set))
    observed = np.logical_and(~np.isin(sources, new_test), ~np.isin(destinations, new_test))
    train_mask = np.logical_and(timestamps <= val_time, observed)
    train_data = Data(sources[train_mask], destinations[train_mask], timestamps[train_mask], edge_idxs[train_mask],
                      labels[train_mask])
    train_node_set = set(train_data.sources).union(train_data.destinations)
    assert len(train_node_set & new_test_node_set) == 0
    new_node_set = node_set - train_node_set             # the val set can indeed contain the new test node
    new_nodes = np.fromiter(new_node_set, dtype=sources.dtype, count=len(new_node_set))
    val_mask = np.logical_and(timestamps <= test_time, timestamps > val_time)
    test_mask = timestamps > test_time
    has_new = np.logical_or(np.isin(sources, new_nodes), np.isin(destinations, new_nodes))

    def split(mask):
        return Data(sources[mask], destinations[mask], timestamps[mask], edge_idxs[mask], labels[mask])

    val_data, test_data = split(val_mask), split(test_mask)
    new_node_val_data = split(np.logical_and(val_mask, has_new))
    new_node_test_data = split(np.logical_and(test_mask, has_new))
    if verbose:
        for nm, d in (("dataset", full_data), ("training dataset", train_data), ("validation dataset", val_data),
                      ("test dataset", test_data), ("new node validation dataset", new_node_val_data),
                      ("new node test dataset", new_node_test_data)):
            print("The %s has %d interactions, involving %d different nodes" % (nm, d.n_interactions, d.n_unique_nodes))
        print("%d nodes were used for the inductive testing, i.e. are never seen during training" % len(new_test_node_set))
    return (full_data, train_data, val_data, test_data, new_node_val_data, new_node_test_data, n_total_unique_nodes,
            n_edges)


def _gaps(ids, timestamps):
    """Per event: time since the previous event of the same id (since 0 for the first)."""
    order = np.argsort(ids, kind="stable")
    t = np.asarray(timestamps)[order]
    prev = np.concatenate([[0], t[:-1]]).astype(t.dtype)
    first = np.ones(len(t), bool)
    first[1:] = np.asarray(ids)[order][1:] != np.asarray(ids)[order][:-1]
    prev[first] = 0
    out = np.empty_like(t)
    out[order] = t - prev
    return out


def compute_time_statistics(sources, destinations, timestamps):
    """utils/data_processing.py:35-64 (mean/std of the inter-event times per source and per destination)."""
    ds, dd = _gaps(sources, timestamps), _gaps(destinations, timestamps)
    return np.mean(ds), np.std(ds), np.mean(dd), np.std(dd)
