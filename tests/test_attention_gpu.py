"""k_temporal_attention against the reference's own TemporalAttentionLayer
(model/temporal_attention.py:7-68): fixture g9_attention_* holds the outputs that class produced for
seeded weights and inputs (tests/golden/gen_golden.py); larger shapes are checked against the numpy
restatement oracle/attention_np.py, itself pinned by the same fixture.  Tolerance 1e-4."""
import os
import sys

import numpy as np
import pytest
import torch

import inputs as I
from conftest import golden, ROOT

pytestmark = pytest.mark.gpu


def _layer(D, F, T, heads, w):
    from zebra_amd.modules import TemporalAttentionLayer
    layer = TemporalAttentionLayer(D, D, F, T, output_dimension=D, n_head=heads, dropout=0.1).eval()
    mha = layer.multi_head_target
    with torch.no_grad():
        for p, kk in ((mha.q_proj_weight, "q_w"), (mha.k_proj_weight, "k_w"), (mha.v_proj_weight, "v_w"),
                      (mha.in_proj_bias, "in_b"), (mha.out_proj.weight, "out_w"), (mha.out_proj.bias, "out_b"),
                      (layer.merger.fc1.weight, "m1_w"), (layer.merger.fc1.bias, "m1_b"),
                      (layer.merger.fc2.weight, "m2_w"), (layer.merger.fc2.bias, "m2_b")):
            p.copy_(torch.from_numpy(w[kk]))
    return layer.cuda()


def _run(layer, x):
    src, src_t, nbr, nbr_t, edge, mask = [torch.from_numpy(a).cuda() for a in x]
    with torch.no_grad():
        out, aw = layer(src, src_t, nbr, nbr_t, edge, mask)
    return out.cpu().numpy(), aw.cpu().numpy()


@pytest.mark.parametrize("name", list(I.ATTENTION_CASES))
def test_temporal_attention_matches_reference_layer(name):
    D, F, T, k, N, heads, seed = I.ATTENTION_CASES[name]
    g = golden("g9_attention_" + name)
    out, aw = _run(_layer(D, F, T, heads, I.attention_weights(D, F, T, seed)), I.attention_inputs(D, F, T, k, N, seed))
    assert out.shape == g["out"].shape and aw.shape == g["attn_w"].shape
    assert np.abs(out - g["out"]).max() <= 1e-4
    assert np.abs(aw - g["attn_w"]).max() <= 1e-5
    assert np.abs(aw[3]).max() == 0.0                  # the row without any neighbour (:58-66)


@pytest.mark.parametrize("D,F,T,k,N,heads", [(100, 172, 100, 20, 600, 2), (100, 1, 100, 40, 257, 2), (20, 4, 12, 5, 70, 4)])
def test_temporal_attention_matches_oracle(D, F, T, k, N, heads):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import attention_np
    w = I.attention_weights(D, F, T, 900 + k)
    x = I.attention_inputs(D, F, T, k, N, 900 + k)
    out, aw = _run(_layer(D, F, T, heads, w), x)
    want_out, want_w = attention_np.temporal_attention(*x, w, heads)
    assert np.abs(out - want_out).max() <= 1e-4
    assert np.abs(aw - want_w).max() <= 1e-5
