"""TemporalAttentionLayer (reference model/temporal_attention.py:7-68).  The layer is dead code in the
reference (no caller, no outputs to capture), so the HIP kernel is checked against the torch restatement
(nn.MultiheadAttention + MergeLayer) it mirrors, within 1e-4."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("D,F,T,k,N,heads", [(100, 172, 100, 20, 600, 2), (100, 1, 100, 10, 257, 2), (20, 4, 12, 5, 70, 4),
                                             (100, 172, 100, 40, 50, 2)])
def test_temporal_attention_matches_torch(D, F, T, k, N, heads):
    from zebra_amd.modules import TemporalAttentionLayer
    torch.manual_seed(7)
    dev = torch.device("cuda")
    layer = TemporalAttentionLayer(D, D, F, T, output_dimension=D, n_head=heads, dropout=0.1).to(dev).eval()
    g = torch.Generator().manual_seed(3)
    src = torch.randn(N, D, generator=g).to(dev)
    src_t = torch.cos(torch.zeros(N, 1, T)).to(dev)
    nf = torch.randn(N, k, D, generator=g).to(dev)
    ef = torch.randn(N, k, F, generator=g).to(dev)
    nt = torch.cos(torch.randn(N, k, T, generator=g) * 5).to(dev)
    mask = (torch.rand(N, k, generator=g) < 0.4).to(dev)
    mask[3] = True                     # a row without any neighbour
    mask[5] = False
    with torch.no_grad():
        want_out, want_w = layer.forward_torch(src, src_t, nf, nt, ef, mask)
        got_out, got_w = layer(src, src_t, nf, nt, ef, mask.clone())
    assert got_out.shape == want_out.shape and got_w.shape == want_w.shape
    assert float((got_out - want_out).abs().max()) <= 1e-4
    assert float((got_w - want_w).abs().max()) <= 1e-5
    assert float(got_w[3].abs().max()) == 0.0
