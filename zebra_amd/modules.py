"""Host-side mirror of the reference's hot-path modules over the HIP library.

Class, method, parameter and attribute names follow the reference so that a
driver written against it (and its ``state_dict``) keeps working:

  TimeEncode                model/time_encoding.py:6-28
  MergeLayer                utils/util.py:14-26
  Memory                    modules/memory.py:7-60
  GRUMemoryUpdater          modules/memory_updater.py:19-98
  GraphDiffusionEmbedding   modules/embedding_module.py:76-336
  TemporalAttentionLayer    model/temporal_attention.py:7-68 (dead code in the reference)

Eval-mode forward (the path the metric is defined on) runs entirely in
libzebra_amd.so.  ``train=True`` needs autograd through the aggregation and
the GRU; that path is composed from torch device ops on the same device-side
T-PPR outputs ("next") -- never from the CPU oracle.
"""
import ctypes as C
import time

import numpy as np
import torch
from torch import nn

from . import _capi
from ._capi import check, lib, ptr, stream_ptr
from .tppr import tppr_finder


class TimeEncode(nn.Module):
    """cos(t * w + 0), w_i = 10^(-9i/(dim-1)) evaluated in float32 by numpy
    exactly as the reference does (model/time_encoding.py:18)."""

    def __init__(self, dimension):
        super().__init__()
        self.dimension = dimension
        self.w = nn.Linear(1, dimension)
        self.reset_parameters()

    def reset_parameters(self):
        self.w.weight = nn.Parameter(
            torch.from_numpy(1 / 10 ** np.linspace(0, 9, self.dimension, dtype=np.float32)).reshape(self.dimension, -1))
        self.w.bias = nn.Parameter(torch.zeros(self.dimension))
        self.w.weight.requires_grad = False
        self.w.bias.requires_grad = False

    @torch.no_grad()
    def forward(self, t):
        return torch.cos(self.w(t.unsqueeze(dim=2)))


class MergeLayer(nn.Module):
    def __init__(self, dim1, dim2, dim3, dim4):
        super().__init__()
        self.fc1 = nn.Linear(dim1 + dim2, dim3)
        self.fc2 = nn.Linear(dim3, dim4)
        self.act = nn.ReLU()
        nn.init.xavier_normal_(self.fc1.weight)
        nn.init.xavier_normal_(self.fc2.weight)

    def forward(self, x1, x2):
        return self.fc2(self.act(self.fc1(torch.cat([x1, x2], dim=1))))


class Memory(nn.Module):
    """Dense per-node state.  Unlike the reference, the pending-message flags
    live on the device (``flags``, uint8); ``nodes`` returns a host bool copy
    for code that inspects them."""

    def __init__(self, n_nodes, memory_dimension, input_dimension, message_dimension=None, device="cuda",
                 combination_method="sum", reference_compat_aliasing=False):
        super().__init__()
        # True: backup_memory hands out the live flag buffer itself, as the reference does with its host
        # array (modules/memory.py:50,53) -- a restore then keeps whatever the flags have become since.
        # False (default): the flags are part of the snapshot.  Fixture g10_epoch records the reference.
        self.reference_compat_aliasing = bool(reference_compat_aliasing)
        self.n_nodes = n_nodes
        self.memory_dimension = memory_dimension
        self.input_dimension = input_dimension
        self.message_dimension = message_dimension
        self.device = torch.device(device)
        self.combination_method = combination_method
        self.__init_memory__()

    def __init_memory__(self):
        d = self.device
        self.memory = torch.zeros((self.n_nodes, self.memory_dimension), device=d)
        self.last_update = torch.zeros(self.n_nodes, device=d)
        self._flag_buf = torch.zeros((self.n_nodes + 3) // 4 * 4, dtype=torch.uint8, device=d)
        self.messages = torch.zeros((self.n_nodes, self.message_dimension), device=d)
        self.timestamps = torch.zeros(self.n_nodes, device=d)

    @property
    def flags(self):
        return self._flag_buf[: self.n_nodes]

    def row_map(self):
        """int32[n_nodes], all -1 between uses: node -> row of a compact overlay (the training forward's
        lazily updated rows, GraphDiffusionEmbedding._train_forward)."""
        rm = getattr(self, "_row_map", None)
        if rm is None or rm.numel() != self.n_nodes or rm.device != self.memory.device:
            rm = self._row_map = torch.full((self.n_nodes,), -1, dtype=torch.int32, device=self.memory.device)
        return rm

    @property
    def nodes(self):
        return self.flags.cpu().numpy().astype(bool)

    def _ids(self, idx):
        if torch.is_tensor(idx):
            return idx.to(self.device).long()
        return torch.as_tensor(np.asarray(idx), device=self.device).long()

    def store_raw_messages(self, nodes, messages, timestamps):
        ids = self._ids(nodes)
        self.flags[ids] = 1
        self.messages[ids] = messages
        self.timestamps[ids] = timestamps

    def get_memory(self, node_idxs):
        return self.memory[self._ids(node_idxs), :]

    def set_memory(self, node_idxs, values):
        self.memory[self._ids(node_idxs), :] = values

    def get_last_update(self, node_idxs):
        return self.last_update[self._ids(node_idxs)]

    def backup_memory(self):                                    # modules/memory.py:49-50
        flags = self._flag_buf if self.reference_compat_aliasing else self._flag_buf.clone()
        return (self.memory.clone(), self.last_update.clone(), self.messages.clone(), flags, self.timestamps.clone())

    def restore_memory(self, memory_backup):                    # modules/memory.py:52-53
        self.memory, self.last_update, self.messages = (memory_backup[0].clone(), memory_backup[1].clone(),
                                                        memory_backup[2].clone())
        self._flag_buf = memory_backup[3] if self.reference_compat_aliasing else memory_backup[3].clone()
        self.timestamps = memory_backup[4].clone()

    def detach_memory(self):
        self.memory.detach_()
        self.messages.detach_()

    def clear_messages(self, positives):
        self.flags[self._ids(positives)] = 0


class GRUMemoryUpdater(nn.Module):
    """SequenceMemoryUpdater + nn.GRUCell (modules/memory_updater.py:19-98)."""

    def __init__(self, message_dimension, memory_dimension, device):
        super().__init__()
        self.layer_norm = nn.LayerNorm(memory_dimension)       # dead parameter in the reference; kept for state_dict
        self.message_dimension = message_dimension
        self.device = torch.device(device)
        self.memory_updater = nn.GRUCell(input_size=message_dimension, hidden_size=memory_dimension)
        self.t_index = self.t_real_update = self.t_others = 0
        self._ws = None

    def _weights(self):
        g = self.memory_updater
        self._gw = _capi.GruWeights(ptr(g.weight_ih.detach()), ptr(g.weight_hh.detach()), ptr(g.bias_ih.detach()),
                                    ptr(g.bias_hh.detach()))
        return self._gw

    def _workspace(self, max_rows, D):
        need = lib().zt_gru_workspace_bytes(C.c_int64(max_rows), C.c_int32(D), C.c_int32(self.message_dimension))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(int(need), dtype=torch.uint8, device=self.device)
        return self._ws

    def update_device(self, memory, ids_d=None, n_ids=0, n_ids_d=None):
        """GRU-update the flagged subset of ``ids_d`` (None = all nodes) in place
        and clear their flags; everything stays on the device."""
        D = memory.memory_dimension
        max_rows = memory.n_nodes if ids_d is None else int(n_ids)
        if max_rows == 0:
            return
        ws = self._workspace(max_rows, D)
        g = self.memory_updater
        # (the packed weights sit before the row list in the workspace: they survive a change of max_rows)
        key = (ws.data_ptr(),) + tuple((t.data_ptr(), t._version) for t in (g.weight_ih, g.weight_hh))
        ready = key == getattr(self, "_ws_key", None)
        check(lib().zt_gru_update(ptr(memory.memory), ptr(memory.last_update), ptr(memory.messages),
                                  ptr(memory.timestamps), ptr(memory._flag_buf), C.c_int64(memory.n_nodes),
                                  C.c_int32(D), C.c_int32(self.message_dimension), ptr(ids_d), C.c_int64(n_ids),
                                  ptr(n_ids_d), C.byref(self._weights()), ptr(ws), C.c_int32(1 if ready else 0),
                                  stream_ptr()), "zt_gru_update")
        self._ws_key = key
        hook = getattr(memory, "_rows_changed", None)
        if hook is not None:                       # e.g. the embedding module's projected table follows the rows
            rows, count = self.last_rows()
            hook(rows, count, max_rows)

    def last_rows(self):
        """(rows int32[max_rows], count int32[1]) views of the ids the last update_device
        call updated (layout of the zt_gru_update workspace: count at byte 0, rows at zt_gru_rows_offset)."""
        D = self.memory_updater.hidden_size
        off = int(lib().zt_gru_rows_offset(C.c_int32(D), C.c_int32(self.message_dimension)))
        return self._ws[off:].view(torch.int32), self._ws[:4].view(torch.int32)

    @torch.no_grad()
    def update_memory(self, memory, positives):                 # modules/memory_updater.py:29-43
        ids = torch.as_tensor(np.ascontiguousarray(positives, np.int32), device=self.device)
        self.update_device(memory, ids, ids.numel())

    @torch.no_grad()
    def update_memory_in_test(self, memory):                    # modules/memory_updater.py:46-57
        self.update_device(memory, None)

    def get_updated_memory(self, memory, index=None):           # modules/memory_updater.py:61-90
        """Train-mode lazily-updated copy (autograd flows through the GRU).
        Device torch ops; the reference clones the whole memory here too."""
        flags = memory.flags
        if index is None:
            ids = torch.nonzero(flags, as_tuple=False).view(-1)
        else:
            idx = torch.as_tensor(np.asarray(index), device=self.device).long()
            ids = idx[flags[idx] != 0]
        if ids.numel() == 0:
            return memory.memory.clone(), memory.last_update.clone()
        updated_memory = memory.memory.clone()
        updated_memory[ids] = self.memory_updater(memory.messages[ids], updated_memory[ids])
        updated_last_update = memory.last_update.clone()
        updated_last_update[ids] = memory.timestamps[ids]
        return updated_memory, updated_last_update


def get_memory_updater(module_type, message_dimension, memory_dimension, device):
    if module_type != "gru":
        raise ValueError("only the GRU memory updater is on the accelerated path (got %r)" % module_type)
    return GRUMemoryUpdater(message_dimension, memory_dimension, device)


def _gemm(a, b, m, n, k, lda, ldb, ta, tb, out=None, accumulate=False):
    """out[m, n] = op(a)[m, k] op(b)[k, n] on f32 MFMA (zt_gemm_f32); a, b contiguous float32 CUDA tensors."""
    if out is None:
        out = torch.empty((m, n), dtype=torch.float32, device=a.device)
    check(lib().zt_gemm_f32(ptr(a), ptr(b), ptr(out), C.c_int64(m), C.c_int64(n), C.c_int64(k), C.c_int64(lda),
                            C.c_int64(ldb), C.c_int64(n), C.c_int32(1 if ta else 0), C.c_int32(1 if tb else 0),
                            C.c_int32(1 if accumulate else 0), stream_ptr()), "zt_gemm_f32")
    return out


class _HipLinear(torch.autograd.Function):
    """y = x W^T (+ b) with forward and backward on the HIP GEMM (csrc/train_ops.hip): the nn.Linear layers that act
    on [N, D] matrices in a training step (fc2, fc1_source, fc2_source; modules/embedding_module.py:86-98)."""

    @staticmethod
    def forward(ctx, x, w, b):
        x = x.contiguous()
        w = w.contiguous()
        n, k = x.shape
        o = w.shape[0]
        y = _gemm(x, w, n, o, k, k, k, False, True)                 # x [n, k] . w[o, k]^T
        if b is not None:
            y += b
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        n, k = x.shape
        o = w.shape[0]
        dx = _gemm(dy, w, n, k, o, o, k, False, False) if ctx.needs_input_grad[0] else None      # dy [n, o] . w [o, k]
        dw = _gemm(dy, x, o, k, n, o, k, True, False) if ctx.needs_input_grad[1] else None       # dy^T [o, n] . x [n, k]
        db = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = torch.empty(o, dtype=torch.float32, device=dy.device)
            check(lib().zt_colsum_f32(ptr(dy), C.c_int64(n), C.c_int64(o), C.c_int64(o), ptr(db), C.c_int32(0), stream_ptr()),
                  "zt_colsum_f32")
        return dx, dw, db


class _HipGruRows(torch.autograd.Function):
    """overlay[u] = GRUCell(messages[ids[u]], memory[ids[u]]) -- the lazily updated rows of get_updated_memory
    (modules/memory_updater.py:61-90) -- with the backward to the four GRU parameters on the HIP kernels
    (zt_gru_train_forward / zt_gru_train_backward).  Messages and memory are buffers: no gradient flows into them."""

    @staticmethod
    def forward(ctx, w_ih, w_hh, b_ih, b_hh, messages, memory_t, ids32):
        U, D, msg = int(ids32.numel()), memory_t.shape[1], messages.shape[1]
        dev = memory_t.device
        h = torch.empty((U, D), dtype=torch.float32, device=dev)
        saved = torch.empty((U, 4 * D), dtype=torch.float32, device=dev)
        ws = torch.empty(int(lib().zt_gru_train_workspace_bytes(C.c_int64(U), C.c_int32(D), C.c_int32(msg))), dtype=torch.uint8,
                         device=dev)
        wt = _capi.GruWeights(ptr(w_ih.detach().contiguous()), ptr(w_hh.detach().contiguous()), ptr(b_ih.detach().contiguous()),
                              ptr(b_hh.detach().contiguous()))
        check(lib().zt_gru_train_forward(ptr(messages), ptr(memory_t), ptr(ids32), C.c_int64(U), C.c_int32(D), C.c_int32(msg),
                                         C.byref(wt), ptr(h), ptr(saved), ptr(ws), stream_ptr()), "zt_gru_train_forward")
        ctx.save_for_backward(saved, ids32)
        ctx.misc = (messages, memory_t, ws, w_ih.shape, w_hh.shape)
        return h

    @staticmethod
    def backward(ctx, dh):
        saved, ids32 = ctx.saved_tensors
        messages, memory_t, ws, s_ih, s_hh = ctx.misc
        U, D, msg = int(ids32.numel()), memory_t.shape[1], messages.shape[1]
        dev = memory_t.device
        d_w_ih = torch.empty(s_ih, dtype=torch.float32, device=dev)
        d_w_hh = torch.empty(s_hh, dtype=torch.float32, device=dev)
        d_b_ih = torch.empty(3 * D, dtype=torch.float32, device=dev)
        d_b_hh = torch.empty(3 * D, dtype=torch.float32, device=dev)
        check(lib().zt_gru_train_backward(ptr(dh.contiguous()), ptr(messages), ptr(memory_t), ptr(ids32), C.c_int64(U),
                                          C.c_int32(D), C.c_int32(msg), ptr(saved), ptr(d_w_ih), ptr(d_w_hh), ptr(d_b_ih),
                                          ptr(d_b_hh), ptr(ws), stream_ptr()), "zt_gru_train_backward")
        return d_w_ih, d_w_hh, d_b_ih, d_b_hh, None, None, None


class _OverlayRows(torch.autograd.Function):
    """rows[i] = get_updated_memory(...)[nodes[i]] (modules/memory_updater.py:61-90, modules/embedding_module.py:320-322):
    the overlay row where ``row_map`` names one, else the memory row; the gradient flows to the overlay rows only (memory is a
    buffer).  One kernel forward, one backward (zt_overlay_rows / zt_overlay_rows_backward) in place of an index, a clamp, a
    compare, a where and -- backward -- a sort-based index_put."""

    @staticmethod
    def forward(ctx, overlay, memory_t, row_map, nodes32, use_map):
        n, D = int(nodes32.numel()), memory_t.shape[1]
        out = torch.empty((n, D), dtype=torch.float32, device=memory_t.device)
        sel = torch.empty(n, dtype=torch.int32, device=memory_t.device)
        ov = overlay.detach().contiguous()
        check(lib().zt_overlay_rows(ptr(memory_t), ptr(ov) if use_map else None, ptr(row_map) if use_map else None, ptr(nodes32),
                                    C.c_int64(n), C.c_int32(D), C.c_int64(memory_t.shape[0]), ptr(out), ptr(sel), stream_ptr()),
              "zt_overlay_rows")
        ctx.save_for_backward(sel)
        ctx.shape = tuple(overlay.shape)
        return out

    @staticmethod
    def backward(ctx, d_out):
        (sel,) = ctx.saved_tensors
        d_ov = torch.zeros(ctx.shape, dtype=torch.float32, device=d_out.device)
        check(lib().zt_overlay_rows_backward(ptr(d_out.contiguous()), ptr(sel), C.c_int64(sel.numel()), C.c_int32(ctx.shape[1]),
                                             ptr(d_ov), stream_ptr()), "zt_overlay_rows_backward")
        return d_ov, None, None, None, None


class _NeighbourAggregate(torch.autograd.Function):
    """H[m][n] = sum_k w_k/sum(w) relu(fc1([memory'[nbr] | ef | cos(dt w)])), S[m][n] = (sum_k w != 0) with
    memory' = overlay rows where row_map says so (zt_agg_train_forward / zt_agg_train_backward)."""

    @staticmethod
    def forward(ctx, overlay, fc1_w, fc1_b, em, memory_t, row_map, ids32, on, oe, od, ow, drop_p=0.0, drop_seed=0):
        M, N, k = on.shape
        D, F, T = em.embedding_dimension, em.n_edge_features, em.n_time_features
        need = lib().zt_embed_workspace_bytes(C.c_int64(N), C.c_int32(D), C.c_int32(F), C.c_int32(T), C.c_int32(M),
                                              C.c_int32(k))
        if need < 0:
            raise ValueError("zt_agg_train_forward: unsupported shape")
        ws = torch.empty(int(need), dtype=torch.uint8, device=on.device)
        H = torch.empty((M, N, D), dtype=torch.float32, device=on.device)
        S = torch.empty((M, N), dtype=torch.float32, device=on.device)
        if em._status is None:
            em._status = torch.zeros(1, dtype=torch.int32, device=on.device)
        ov = overlay.detach().contiguous()
        use_map = ids32 is not None
        check(lib().zt_agg_train_forward(ptr(memory_t), ptr(ov) if use_map else None, ptr(row_map) if use_map else None,
                                         ptr(em.edge_features), C.c_int64(memory_t.shape[0]),
                                         C.c_int64(em.edge_features.shape[0]), C.c_int32(D), C.c_int32(F), C.c_int32(T),
                                         C.c_int64(N), C.c_int32(M), C.c_int32(k), ptr(on), ptr(oe), ptr(od), ptr(ow),
                                         C.byref(em._embed_weights()), ptr(H), ptr(S), ptr(ws), ptr(em._status),
                                         C.c_float(drop_p), C.c_uint64(drop_seed), stream_ptr()), "zt_agg_train_forward")
        ctx.save_for_backward(ov, fc1_w, fc1_b)
        ctx.misc = (em, memory_t, row_map, ids32, on, oe, od, ow, float(drop_p), int(drop_seed))
        ctx.mark_non_differentiable(S)
        return H, S

    @staticmethod
    def backward(ctx, dH, dS):
        ov, fc1_w, fc1_b = ctx.saved_tensors
        em, memory_t, row_map, ids32, on, oe, od, ow, drop_p, drop_seed = ctx.misc
        M, N, k = on.shape
        D, F, T = em.embedding_dimension, em.n_edge_features, em.n_time_features
        dW1 = torch.zeros_like(fc1_w)
        db1 = torch.zeros_like(fc1_b)
        d_ov = torch.zeros_like(ov)
        use_map = ids32 is not None
        if use_map:                                       # the map is shared scratch: set for this call, reset after
            row_map[ids32.long()] = torch.arange(ids32.numel(), dtype=torch.int32, device=ov.device)
        ws = torch.empty(int(lib().zt_agg_backward_workspace_bytes(C.c_int32(D), C.c_int32(F), C.c_int32(T))),
                         dtype=torch.uint8, device=ov.device)
        w1 = fc1_w.detach().contiguous()
        check(lib().zt_agg_train_backward(ptr(memory_t), ptr(ov) if use_map else None, ptr(row_map) if use_map else None,
                                          ptr(em.edge_features), ptr(em.time_encoder.w.weight),
                                          C.c_int64(memory_t.shape[0]), C.c_int64(em.edge_features.shape[0]), C.c_int32(D),
                                          C.c_int32(F), C.c_int32(T), C.c_int64(N), C.c_int32(M), C.c_int32(k), ptr(on),
                                          ptr(oe), ptr(od), ptr(ow), ptr(w1), ptr(fc1_b.detach()),
                                          ptr(dH.contiguous()), ptr(dW1), ptr(db1), ptr(d_ov) if use_map else None,
                                          ptr(ws), C.c_float(drop_p), C.c_uint64(drop_seed), stream_ptr()),
              "zt_agg_train_backward")
        if use_map:
            row_map[ids32.long()] = -1
        return d_ov, dW1, db1, None, None, None, None, None, None, None, None, None, None


def dropout_mask(seed, p, shape_mnk, D):
    """The keep-mask the fused training kernels derive from (seed, element) -- csrc/common.hpp drop_scale in numpy:
    float32 [M, N, k, D] of 1 / (1 - p) and 0.  For tests (the kernels never materialise it)."""
    M, N, k = shape_mnk
    idx = np.arange(M * N * k * D, dtype=np.uint64)
    lo, hi = np.uint32(seed & 0xFFFFFFFF), np.uint32((seed >> 32) & 0xFFFFFFFF)
    with np.errstate(over="ignore"):
        h = ((idx & np.uint64(0xFFFFFFFF)).astype(np.uint32) * np.uint32(0x9E3779B1)) ^ lo
        h = h ^ ((idx >> np.uint64(32)).astype(np.uint32) * np.uint32(0x85EBCA77) + hi)
        h ^= h >> np.uint32(16); h *= np.uint32(0x85EBCA6B); h ^= h >> np.uint32(13); h *= np.uint32(0xC2B2AE35); h ^= h >> np.uint32(16)
    t = float(np.float32(p)) * 4294967296.0
    thr = np.uint32(4294967295 if t >= 4294967295.0 else int(t))
    keep = h >= thr
    return (keep.astype(np.float32) * np.float32(1.0 / (1.0 - float(np.float32(p))))).reshape(M, N, k, D)


class GraphDiffusionEmbedding(nn.Module):
    """modules/embedding_module.py:76-336 (GraphEmbedding -> GraphDiffusionEmbedding)."""

    def __init__(self, node_features, edge_features, memory, neighbor_finder, time_encoder, n_layers,
                 n_node_features, n_edge_features, n_time_features, embedding_dimension, device, n_heads=2,
                 dropout=0.1, use_memory=True, args=None, num_nodes=-1):
        super().__init__()
        self.node_features = node_features
        self.edge_features = edge_features
        self.neighbor_finder = neighbor_finder
        self.time_encoder = time_encoder
        self.n_layers = n_layers
        self.n_node_features = n_node_features
        self.n_edge_features = n_edge_features
        self.n_time_features = n_time_features
        self.dropout = dropout
        self.embedding_dimension = embedding_dimension
        self.device = torch.device(device)
        self.use_memory = use_memory
        self.args = args
        self.num_nodes = num_nodes
        self.t_tppr = 0
        self.sync_timers = False

        self.fc1 = nn.Linear(embedding_dimension + n_time_features + n_edge_features, embedding_dimension)
        self.fc2 = nn.Linear(embedding_dimension, embedding_dimension)
        self.act = nn.ReLU()
        self.drop = nn.Dropout(0.1)
        nn.init.xavier_normal_(self.fc1.weight)
        nn.init.xavier_normal_(self.fc2.weight)
        self.fc1_source = nn.Linear(embedding_dimension, embedding_dimension)
        self.fc2_source = nn.Linear(embedding_dimension, embedding_dimension)
        nn.init.xavier_normal_(self.fc1_source.weight)
        nn.init.xavier_normal_(self.fc2_source.weight)
        self.combiner = nn.Linear(embedding_dimension + embedding_dimension, embedding_dimension)   # dead, as in the reference

        self.n_tppr = len(args.alpha_list)
        self.alpha_list = list(args.alpha_list)
        self.beta_list = list(args.beta_list)
        self.k = args.topk
        self.tppr_strategy = args.tppr_strategy
        self.width = args.n_degree
        self.depth = args.n_layer
        assert self.k != 0
        if self.tppr_strategy == "streaming":
            self.tppr_finder = tppr_finder(self.num_nodes, self.k, self.n_tppr, self.alpha_list, self.beta_list,
                                           reference_compat_aliasing=getattr(args, "reference_compat_aliasing", False))
        self._ws = None
        self._ws_key = None                     # weights the workspace's padded copies were made from
        self._ws_shape = None
        self._proj = None                       # projected memory table (see embed_device)
        self.use_projection = getattr(args, "use_projection", True)
        self._status = None
        self._avg_topk_t = None

    # ---- T-PPR state management (modules/embedding_module.py:114-136) ----
    def reset_tppr(self):
        self.tppr_finder.reset_tppr()

    def backup_tppr(self):
        return self.tppr_finder.backup_tppr()

    def restore_tppr(self, backup):
        self.tppr_finder.restore_tppr(backup)

    def streaming_topk(self, source_nodes, timestamps, edge_idxs):
        return self.tppr_finder.streaming_topk(source_nodes, timestamps, edge_idxs)

    def streaming_topk_no_fake(self, source_nodes, timestamps, edge_idxs):
        return self.tppr_finder.streaming_topk_no_fake(source_nodes, timestamps, edge_idxs)

    def fill_tppr(self, sources, targets, timestamps, edge_idxs, tppr_filled):
        if tppr_filled:
            self.tppr_finder.restore_val_tppr()
        else:
            self.tppr_finder.compute_val_tppr(sources, targets, timestamps, edge_idxs)

    def pruning_topk(self, source_nodes, timestamps):           # modules/embedding_module.py:280-297
        on, oe, od, ow = self.pruning_topk_device(
            torch.as_tensor(np.ascontiguousarray(source_nodes, np.int32), device=self.device),
            torch.as_tensor(np.ascontiguousarray(timestamps, np.float64), device=self.device))
        return ([a for a in on.cpu().numpy()], [a for a in oe.cpu().numpy()], [a for a in od.cpu().numpy()],
                [a for a in ow.cpu().numpy()])

    def pruning_topk_device(self, nodes_d, ts_d, check_status=True):
        n = nodes_d.numel()
        on = torch.zeros((self.n_tppr, n, self.k), dtype=torch.int32, device=self.device)
        oe = torch.zeros_like(on)
        od = torch.zeros((self.n_tppr, n, self.k), dtype=torch.float32, device=self.device)
        ow = torch.zeros_like(od)
        self.neighbor_finder.pruned_topk_multi_device(nodes_d, ts_d, self.width, self.depth, self.alpha_list,
                                                      self.beta_list, self.k, on, oe, od, ow, check_status=check_status)
        return on, oe, od, ow

    @property
    def average_topk(self):
        return float(self._avg_topk_t.item()) if self._avg_topk_t is not None else 0.0

    # ---- the hot path ----
    def _embed_weights(self):
        p = lambda t: ptr(t.detach())
        self._ew = _capi.EmbedWeights(p(self.fc1.weight), p(self.fc1.bias), p(self.fc2.weight), p(self.fc2.bias),
                                      p(self.fc1_source.weight), p(self.fc1_source.bias), p(self.fc2_source.weight),
                                      p(self.fc2_source.bias), p(self.time_encoder.w.weight))
        return self._ew

    def topk_device(self, nodes_d, ts_d, eidx_d, check_status=True, plan_token=0):
        """T-PPR query for one batch, device tensors in and out."""
        if self.tppr_strategy == "streaming":
            B = eidx_d.numel()
            return self.tppr_finder.stream_device(nodes_d, ts_d[:B].contiguous(), eidx_d, 3 if nodes_d.numel() == 3 * B
                                                  else 2, True, -1, check_status=check_status, plan_token=plan_token)
        if self.tppr_strategy == "pruning":
            return self.pruning_topk_device(nodes_d, ts_d, check_status=check_status)
        raise ValueError("tppr_strategy must be 'streaming' or 'pruning'")

    def _weights_key(self):
        """Identity + in-place version of every weight the kernels read: the padded copies in the workspace are
        remade only when this changes (an optimizer step, load_state_dict, .to())."""
        ps = (self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, self.fc1_source.weight,
              self.fc1_source.bias, self.fc2_source.weight, self.fc2_source.bias, self.time_encoder.w.weight)
        return tuple((t.data_ptr(), t._version) for t in ps)

    def _workspace(self, N):
        """(workspace tensor, weights_ready) for N output rows."""
        D, F, T = self.embedding_dimension, self.n_edge_features, self.n_time_features
        need = lib().zt_embed_workspace_bytes(C.c_int64(N), C.c_int32(D), C.c_int32(F), C.c_int32(T),
                                              C.c_int32(self.n_tppr), C.c_int32(self.k))
        if need < 0:
            raise ValueError("zt_embed: unsupported shape D=%d F=%d T=%d k=%d" % (D, F, T, self.k))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(int(need), dtype=torch.uint8, device=self.device)
            self._ws_key = None
        key = self._weights_key()
        ready = self._ws_key == key
        return self._ws, ready, key

    def _project(self, memory_obj, rows=None, count=None, max_rows=0):
        """Refresh the projected table P[v] = W_m memory[v] for all nodes (rows None) or the listed rows."""
        D, F, T = self.embedding_dimension, self.n_edge_features, self.n_time_features
        ws, ready, key = self._workspace(max(1, self._ws_shape or 1))
        pr = self._proj
        check(lib().zt_project_memory(ptr(memory_obj.memory), C.c_int64(memory_obj.n_nodes), C.c_int32(D), C.c_int32(F),
                                      C.c_int32(T), C.byref(self._embed_weights()), C.c_int32(1 if ready else 0),
                                      ptr(rows), ptr(count), C.c_int64(max_rows), ptr(pr["table"]), ptr(ws),
                                      C.c_int64(max(1, self._ws_shape or 1)), C.c_int32(self.n_tppr), C.c_int32(self.k),
                                      stream_ptr()), "zt_project_memory")
        self._ws_key = key

    def invalidate_projection(self):
        """For callers that write the memory table or fc1's weights behind torch's back (raw pointers): the
        projected table is rebuilt at its next use."""
        self._proj_serial = getattr(self, "_proj_serial", 0) + 1      # part of TGN._pipe_signature
        if self._proj is not None:
            self._proj["key"] = None

    def _projection(self, memory_obj):
        """The projected table for ``memory_obj.memory`` if it can be used, else None.  Valid for one memory tensor
        (identity and torch in-place version: HIP kernels that rewrite rows report them through
        ``memory_obj._rows_changed`` instead) and one set of weights; anything else triggers a full rebuild."""
        if not self.use_projection or memory_obj is None:
            return None
        mem = memory_obj.memory
        key = (mem.data_ptr(), mem._version, self._weights_key())
        pr = self._proj
        if pr is None or pr["table"].shape[0] != memory_obj.n_nodes:
            nbytes = lib().zt_project_table_bytes(C.c_int64(memory_obj.n_nodes), C.c_int32(self.embedding_dimension))
            pr = self._proj = {"table": torch.empty((memory_obj.n_nodes, int(nbytes) // (4 * memory_obj.n_nodes)),
                                                    dtype=torch.float32, device=self.device), "key": None}
        if pr["key"] != key:
            self._project(memory_obj)                                  # every node
            pr["key"] = key
        em = self

        def rows_changed(rows, count, max_rows):                       # called after HIP kernels rewrote memory rows
            p2 = em._proj
            m2 = memory_obj.memory
            if p2 is None or p2["key"] != (m2.data_ptr(), m2._version, em._weights_key()):
                if p2 is not None:
                    p2["key"] = None                                   # stale: rebuilt at the next use
                return
            em._project(memory_obj, rows, count, max_rows)
        memory_obj._rows_changed = rows_changed
        return pr["table"]

    def embed_device(self, memory_t, nodes_d, on, oe, od, ow, check_status=True, memory_obj=None):
        """Eval forward of :243-276 on the HIP kernels: [N, D*(n_tppr+1)].  ``memory_obj``: the Memory whose
        ``.memory`` tensor ``memory_t`` is -- its projected table then replaces fc1's memory columns."""
        N, D = nodes_d.numel(), self.embedding_dimension
        F, T = self.n_edge_features, self.n_time_features
        self._ws_shape = max(N, self._ws_shape or 0)
        table = self._projection(memory_obj) if (memory_obj is not None and memory_t is memory_obj.memory) else None
        ws, ready, key = self._workspace(self._ws_shape)
        if self._status is None:
            self._status = torch.zeros(1, dtype=torch.int32, device=self.device)
        out = torch.empty((N, D * (self.n_tppr + 1)), dtype=torch.float32, device=self.device)
        check(lib().zt_embed(ptr(memory_t), ptr(self.edge_features), C.c_int64(memory_t.shape[0]),
                             C.c_int64(self.edge_features.shape[0]), C.c_int32(D), C.c_int32(F), C.c_int32(T),
                             ptr(nodes_d), C.c_int64(N), C.c_int32(self.n_tppr), C.c_int32(self.k), ptr(on), ptr(oe),
                             ptr(od), ptr(ow), C.byref(self._embed_weights()), ptr(out), ptr(ws),
                             ptr(self._status), ptr(table), C.c_int32(1 if ready else 0), stream_ptr()), "zt_embed")
        self._ws_key = key
        if check_status:
            st = int(self._status.item())
            if st != 0:
                self._status.zero_()
                raise IndexError("zt_embed: node / edge id out of range (status %d)" % st)
        return out

    def compute_embedding_tppr_ensemble(self, memory, source_nodes, timestamps, edge_idxs, memory_updater, train,
                                        row_sel=None, on_device=None):
        """modules/embedding_module.py:217-278.  ``memory`` is the Memory object
        in train mode and the raw memory tensor in eval mode, as in the reference.
        ``row_sel`` (train mode, optional): LongTensor of the rows to embed -- a data-parallel rank embeds (and
        back-propagates through) its share of the batch only; the T-PPR update always covers the whole batch."""
        d = self.device
        if on_device is not None:                  # (the caller has the three arrays on the device already: TGN's training step)
            nodes_d, ts_d, eidx_d = on_device
        else:
            nodes_d, ts_d, eidx_d = _capi.to_device(d, [np.ascontiguousarray(source_nodes, np.int32),
                                                        np.ascontiguousarray(timestamps, np.float64),
                                                        np.ascontiguousarray(edge_idxs, np.int64)])
        t = time.time()
        on, oe, od, ow = self.topk_device(nodes_d, ts_d, eidx_d)
        if self.sync_timers:
            torch.cuda.synchronize()
        self.t_tppr += time.time() - t
        n_edge = ow.shape[1] // 3
        self._avg_topk_t = ow[0, : 2 * n_edge].sum(dim=1).mean()
        if not train:
            with torch.no_grad():
                return self.embed_device(memory, nodes_d, on, oe, od, ow)
        index = None
        if row_sel is not None:
            # which nodes read their lazily updated row is decided by the WHOLE batch's neighbours (:227-230), a
            # node of this share may be another share's neighbour
            index = torch.unique(on.reshape(-1).long())
            nodes_d = nodes_d[row_sel].contiguous()
            on, oe, od, ow = [t[:, row_sel].contiguous() for t in (on, oe, od, ow)]
        return self._train_forward(memory, nodes_d, on, oe, od, ow, memory_updater, index)

    def _train_forward(self, memory, nodes_d, on, oe, od, ow, memory_updater, index=None):
        """Autograd path (:227-276).  The lazily updated memory (get_updated_memory, which clones the whole
        [N, D] table in the reference) is a compact OVERLAY here: the GRU runs on the flagged neighbour rows
        only ([U, msg] x plain GEMMs, autograd to the GRU weights) and ``row_map`` says which nodes read their
        row from it.  The neighbour half -- gather, TimeEncode, fc1, ReLU, weighted k-reduction -- is one
        fused HIP kernel forward and one backward (csrc/aggregate_bwd.hip); fc2 and the source transform act
        on [N, D] matrices."""
        if index is None:
            index = torch.unique(on.reshape(-1).long())
        ids = index[memory.flags[index] != 0]                                   # neighbours with a pending message
        U = int(ids.numel())
        hip_dense = getattr(self, "fused_training", True)
        if U and hip_dense:
            g = memory_updater.memory_updater                                   # nn.GRUCell's parameters, HIP forward + backward
            overlay = _HipGruRows.apply(g.weight_ih, g.weight_hh, g.bias_ih, g.bias_hh, memory.messages, memory.memory,
                                        ids.to(torch.int32).contiguous())      # [U, D]
        elif U:
            overlay = memory_updater.memory_updater(memory.messages[ids], memory.memory[ids])      # [U, D]
        else:
            overlay = torch.zeros((1, self.embedding_dimension), device=self.device)
        row_map = memory.row_map()
        ids32 = ids.to(torch.int32)
        if U:
            row_map[ids] = torch.arange(U, dtype=torch.int32, device=self.device)
        if hip_dense and getattr(self, "overlay_rows_op", True):
            src_rows = _OverlayRows.apply(overlay, memory.memory, row_map, nodes_d.to(torch.int32).contiguous(), bool(U))
        else:
            src_rows = memory.memory[nodes_d.long()]
            if U:
                m = row_map[nodes_d.long()].long()
                src_rows = torch.where((m >= 0).unsqueeze(1), overlay[m.clamp(min=0)], src_rows)
        if hip_dense:                                                           # transform_source on the HIP GEMM (:320-322)
            embeddings = _HipLinear.apply(self.drop(self.act(_HipLinear.apply(src_rows, self.fc1_source.weight, self.fc1_source.bias))),
                                          self.fc2_source.weight, self.fc2_source.bias)
        else:
            embeddings = self.transform_source(src_rows)
        fused = getattr(self, "fused_training", True) and self.embedding_dimension <= 128
        if fused:
            # the dropout of the hidden layer (self.drop, active in train mode) runs inside the kernels: a seed from
            # torch's CPU generator (reproducible under torch.manual_seed), the mask is never materialised
            drop_p = float(self.drop.p) if self.training else 0.0
            drop_seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if drop_p > 0 else 0
            self._last_drop_seed = drop_seed
            H, S = _NeighbourAggregate.apply(overlay, self.fc1.weight, self.fc1.bias, self, memory.memory, row_map,
                                             ids32 if U else None, on, oe, od, ow, drop_p, drop_seed)
            if U:
                row_map[ids] = -1
            for m_ in range(self.n_tppr):
                # fc2 is linear: sum_k w_k fc2(h_k) = fc2(sum_k w_k h_k) + b2 * [sum_k w_k != 0]
                out = _HipLinear.apply(H[m_], self.fc2.weight, None) + self.fc2.bias * S[m_].unsqueeze(1)
                embeddings = torch.cat((embeddings, out), dim=1)
            return embeddings
        # dropout inside the neighbour transform: composed from torch ops on the same overlay
        mem_rows = lambda idx: memory.memory[idx] if not U else torch.where(
            (row_map[idx] >= 0).unsqueeze(-1), overlay[row_map[idx].long().clamp(min=0)], memory.memory[idx])
        for m_ in range(self.n_tppr):
            x = torch.cat([mem_rows(on[m_].long()), self.edge_features[oe[m_].long(), :], self.time_encoder(od[m_])], dim=-1)
            x = self.transform(x)
            w = ow[m_]
            ws = torch.sum(w, dim=1)
            w = w / ws.unsqueeze(1)
            w[ws == 0] = 0
            embeddings = torch.cat((embeddings, torch.sum(x * w[:, :, None], dim=1)), dim=1)
        if U:
            row_map[ids] = -1
        return embeddings

    def transform_source(self, x):
        return self.fc2_source(self.drop(self.act(self.fc1_source(x))))

    def transform(self, x):
        return self.fc2(self.drop(self.act(self.fc1(x))))

    def combine(self, x):
        return self.combiner(x)


class TemporalAttentionLayer(nn.Module):
    """model/temporal_attention.py:7-68.  Dead code in the reference (only GraphAttentionEmbedding builds it
    and train.py can never reach that class); provided because the north star names it.
    Without grad (eval) the forward runs in libzebra_amd.so; with grad it is the torch restatement below,
    which is also what the HIP kernel is tested against (no reference outputs exist)."""

    def __init__(self, n_node_features, n_neighbors_features, n_edge_features, time_dim, output_dimension,
                 n_head=2, dropout=0.1):
        super().__init__()
        self.n_head = n_head
        self.feat_dim = n_node_features
        self.time_dim = time_dim
        self.query_dim = n_node_features + time_dim
        self.key_dim = n_neighbors_features + time_dim + n_edge_features
        self.n_edge_features = n_edge_features
        self.output_dimension = output_dimension
        self.merger = MergeLayer(self.query_dim, n_node_features, n_node_features, output_dimension)
        self.multi_head_target = nn.MultiheadAttention(embed_dim=self.query_dim, kdim=self.key_dim,
                                                       vdim=self.key_dim, num_heads=n_head, dropout=dropout)
        self._ws = None

    def forward_torch(self, src_node_features, src_time_features, neighbors_features, neighbors_time_features,
                      edge_features, neighbors_padding_mask):
        query = torch.cat([src_node_features.unsqueeze(1), src_time_features], dim=2).permute([1, 0, 2])
        key = torch.cat([neighbors_features, edge_features, neighbors_time_features], dim=2).permute([1, 0, 2])
        mask = neighbors_padding_mask.clone()
        invalid = mask.all(dim=1, keepdim=True)
        mask[invalid.squeeze(1), 0] = False
        out, w = self.multi_head_target(query=query, key=key, value=key, key_padding_mask=mask)
        out = out.squeeze(0).masked_fill(invalid, 0)
        w = w.squeeze(1).masked_fill(invalid, 0)
        return self.merger(out, src_node_features), w

    def forward(self, src_node_features, src_time_features, neighbors_features, neighbors_time_features,
                edge_features, neighbors_padding_mask):
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()) and self.training:
            return self.forward_torch(src_node_features, src_time_features, neighbors_features,
                                      neighbors_time_features, edge_features, neighbors_padding_mask)
        N, k, D = neighbors_features.shape
        T, F = self.time_dim, edge_features.shape[2]
        E = self.query_dim
        need = lib().zt_attention_workspace_bytes(C.c_int32(D), C.c_int32(F), C.c_int32(T), C.c_int32(self.n_head),
                                                  C.c_int32(self.feat_dim), C.c_int32(self.output_dimension),
                                                  C.c_int32(k))
        if need < 0:
            raise ValueError("zt_temporal_attention: unsupported shape")
        dev = src_node_features.device
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(int(need), dtype=torch.uint8, device=dev)
        mha = self.multi_head_target
        c = lambda t: t.detach().contiguous().float()
        ten = [c(src_node_features), c(src_time_features.reshape(N, T)), c(neighbors_features), c(edge_features),
               c(neighbors_time_features), neighbors_padding_mask.detach().to(torch.uint8).contiguous()]
        wts = [c(mha.q_proj_weight), c(mha.k_proj_weight), c(mha.v_proj_weight), c(mha.in_proj_bias),
               c(mha.out_proj.weight), c(mha.out_proj.bias), c(self.merger.fc1.weight), c(self.merger.fc1.bias),
               c(self.merger.fc2.weight), c(self.merger.fc2.bias)]
        aw = _capi.AttnWeights(*[ptr(t) for t in wts])
        out = torch.empty((N, self.output_dimension), dtype=torch.float32, device=dev)
        attn_w = torch.empty((N, k), dtype=torch.float32, device=dev)
        check(lib().zt_temporal_attention(ptr(ten[0]), ptr(ten[1]), ptr(ten[2]), ptr(ten[3]), ptr(ten[4]), ptr(ten[5]),
                                          C.c_int64(N), C.c_int32(k), C.c_int32(D), C.c_int32(F), C.c_int32(T),
                                          C.c_int32(self.n_head), C.c_int32(self.feat_dim),
                                          C.c_int32(self.output_dimension), C.byref(aw), ptr(out), ptr(attn_w),
                                          ptr(self._ws), stream_ptr()), "zt_temporal_attention")
        return out, attn_w


def get_embedding_module(module_type, **kw):
    if module_type != "diffusion":
        raise ValueError("only the 'diffusion' embedding module is reachable from the reference's train.py "
                         "(SURVEY.md 0.1); got %r" % module_type)
    kw.pop("n_neighbors", None)
    return GraphDiffusionEmbedding(**kw)
