"""zebra_amd.data against the reference's get_data / compute_time_statistics
(utils/data_processing.py:35-149), pinned by the fixture g7_ingest (generated from the reference by
tests/golden/gen_golden.py on the synthetic ml_synth.csv of tests/golden/inputs.py)."""
import os

import numpy as np

import inputs as I
from conftest import golden


def test_get_data_matches_reference(tmp_path):
    from zebra_amd import data as zd
    g = golden("g7_ingest")
    u, i, ts, label, idx = I.make_ml_table()
    for a, b in zip((u, i, ts, label, idx), (g["u"], g["i"], g["ts"], g["label"], g["idx"])):
        assert np.array_equal(a, b)                      # the fixture's inputs are the seeded table
    os.makedirs(tmp_path / "synth")
    I.write_ml_csv(tmp_path / "synth" / "ml_synth.csv", u, i, ts, label, idx)
    full, train, val, test, nn_val, nn_test, n_nodes, n_edges = zd.get_data("synth", root=str(tmp_path))
    assert n_nodes == int(g["n_nodes"]) and n_edges == int(g["n_edges"])
    for nm, d in (("full", full), ("train", train), ("val", val), ("test", test), ("nn_val", nn_val), ("nn_test", nn_test)):
        assert np.array_equal(d.edge_idxs, g[nm + "_idx"]), nm
        assert d.n_unique_nodes == int(g[nm + "_n_unique"]), nm
        sel = d.edge_idxs - 1                            # idx is 1-based and the table is in idx order
        assert np.array_equal(d.sources, u[sel]) and np.array_equal(d.destinations, i[sel])
        assert np.array_equal(d.timestamps, ts[sel]) and np.array_equal(d.labels, label[sel])
    got = np.asarray(zd.compute_time_statistics(full.sources, full.destinations, full.timestamps), np.float64)
    assert np.allclose(got, g["time_stats"], rtol=1e-12, atol=0)


def test_load_feat_and_missing_files(tmp_path):
    from zebra_amd import data as zd
    os.makedirs(tmp_path / "x")
    assert zd.load_feat("x", root=str(tmp_path)) == (None, None)
    ef = np.zeros((5, 3), np.float32)
    np.save(tmp_path / "x" / "ml_x.npy", ef)
    nf, got = zd.load_feat("x", root=str(tmp_path))
    assert nf is None and np.array_equal(got, ef)
