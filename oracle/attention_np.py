"""numpy restatement of the reference's TemporalAttentionLayer.forward
(model/temporal_attention.py:35-68): torch.nn.MultiheadAttention with separate
q/k/v projection weights (kdim = vdim != embed_dim), one query token, a
key-padding mask, head-averaged attention weights, then utils/util.py:14-26
MergeLayer.  float64 accumulation, float32 result.

TEST INFRASTRUCTURE ONLY (oracle): pinned by tests/golden/g9_attention_*.npz,
which the reference's own class produced (tests/golden/gen_golden.py).
"""
import numpy as np


def temporal_attention(src, src_t, nbr, nbr_t, edge, mask, w, n_head):
    """src [N,D], src_t [N,1,T], nbr [N,k,D], nbr_t [N,k,T], edge [N,k,F], mask bool [N,k] (True = padding);
    w = dict(q_w, k_w, v_w, in_b, out_w, out_b, m1_w, m1_b, m2_w, m2_b) in torch layout.
    Returns (out [N,Dout], attn_w [N,k])."""
    f8 = np.float64
    N, k, _ = nbr.shape
    query = np.concatenate([src[:, None, :], src_t], axis=2).astype(f8)[:, 0, :]        # :51-52  [N,E]
    key = np.concatenate([nbr, edge, nbr_t], axis=2).astype(f8)                          # :53     [N,k,K]
    E = query.shape[1]
    hd = E // n_head
    mask = np.array(mask, bool, copy=True)
    invalid = mask.all(axis=1)                                                           # :58
    mask[invalid, 0] = False                                                             # :59
    b = w["in_b"].astype(f8)
    q = query @ w["q_w"].astype(f8).T + b[:E]
    kk = key @ w["k_w"].astype(f8).T + b[E:2 * E]
    v = key @ w["v_w"].astype(f8).T + b[2 * E:]
    q = q.reshape(N, n_head, hd) / np.sqrt(hd)
    kk = kk.reshape(N, k, n_head, hd)
    v = v.reshape(N, k, n_head, hd)
    s = np.einsum("nhd,nkhd->nhk", q, kk)
    s = np.where(mask[:, None, :], -np.inf, s)
    s = s - s.max(axis=2, keepdims=True)
    p = np.exp(s)
    p /= p.sum(axis=2, keepdims=True)
    o = np.einsum("nhk,nkhd->nhd", p, v).reshape(N, E)
    o = o @ w["out_w"].astype(f8).T + w["out_b"].astype(f8)
    aw = p.mean(axis=1)                                                                  # head-averaged weights
    o[invalid] = 0                                                                       # :65
    aw[invalid] = 0                                                                      # :66
    h = np.concatenate([o, src.astype(f8)], axis=1) @ w["m1_w"].astype(f8).T + w["m1_b"].astype(f8)
    out = np.maximum(h, 0) @ w["m2_w"].astype(f8).T + w["m2_b"].astype(f8)              # :67, util.py:23-26
    return out.astype(np.float32), aw.astype(np.float32)
