"""SURVEY.md 8f-3 end to end on the device: ml_{name}.csv -> zebra_amd.data.get_data splits ->
get_neighbor_finder (device CSR) -> get_pruned_topk, against the reference's splits (g7_ingest) and the
oracle's adjacency / pruning on the same table (utils/data_processing.py:80-149, utils/util.py:90-107,185-276)."""
import os

import numpy as np
import pytest

import inputs as I
from conftest import golden

pytestmark = pytest.mark.gpu


def test_csv_to_device_csr_to_pruned_topk(tmp_path, oracle):
    from zebra_amd import data as zd
    from zebra_amd.tppr import get_neighbor_finder
    g = golden("g7_ingest")
    u, i, ts, label, idx = I.make_ml_table()
    os.makedirs(tmp_path / "synth")
    I.write_ml_csv(tmp_path / "synth" / "ml_synth.csv", u, i, ts, label, idx)
    full, train, val, test, nn_val, nn_test, n_nodes, n_edges = zd.get_data("synth", root=str(tmp_path))
    assert np.array_equal(train.edge_idxs, g["train_idx"]) and np.array_equal(full.edge_idxs, g["full_idx"])
    for data in (train, full):                             # train_ngh_finder / full_ngh_finder (train.py:138-139)
        nf = get_neighbor_finder(data)
        ref = oracle.CsrOracle(data.sources, data.destinations, data.edge_idxs, data.timestamps)
        assert nf.num_nodes == ref.num_nodes
        assert np.array_equal(nf._indptr, ref.indptr) and np.array_equal(nf._nbr, ref.nbr)
        assert np.array_equal(nf._eid, ref.eid) and np.array_equal(nf._ts, ref.ts)
        for v in (1, 5, int(data.destinations[0])):        # duplicate timestamps: stable order (utils/util.py:99-107)
            t = float(np.median(data.timestamps))
            for a, b in zip(nf.find_before(v, t), ref.find_before(v, t)):
                assert np.array_equal(a, b)
        # queries shaped like a validation batch
        qn = np.concatenate([val.sources[:60], val.destinations[:60]]).astype(np.int32)
        qn = np.minimum(qn, nf.num_nodes - 1)
        qt = np.concatenate([val.timestamps[:60]] * 2)
        k = 10
        got = [np.zeros((len(qn), k), dt) for dt in (np.int32, np.int32, np.float32, np.float32)]
        want = [np.zeros((len(qn), k), dt) for dt in (np.int32, np.int32, np.float32, np.float32)]
        nf.get_pruned_topk(qn, qt, 5, 2, 0.1, 0.5, k, *got)
        ref.get_pruned_topk(qn, qt, 5, 2, 0.1, 0.5, k, *want)
        for a, b in zip(got, want):
            assert np.array_equal(a, b)
        assert (got[3].sum(axis=1) > 0).any()
