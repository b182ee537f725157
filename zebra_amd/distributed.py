"""One-node multi-GPU execution of the eval-mode step.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).
The reference has no distributed code at all; this is new work:

  * T-PPR state, node memory, pending messages and the edge-feature table are
    replicated (C5: ~26 GB of 288 GB per GPU);
  * P1, streaming strategy (T-PPR update + row emission) is applied to the
    WHOLE batch on every rank: the stream is sequential by construction, so
    replicas stay bit-identical with zero communication; P1, pruning strategy
    (a query of the static adjacency, rows independent) is sharded with P2;
  * P2 (gather + aggregate) is sharded over contiguous slices of the 3B rows;
  * P3 (last-message store + GRU update) is sharded by batch POSITION: the
    last occurrence of every endpoint is resolved over the whole batch, rank r
    owns the winners at positions [2B*r/W, 2B*(r+1)/W) -- at most one winner
    per position, so a rank touches at most ceil(2B/W) rows -- and the touched
    rows [id | memory | last_update | message | msg timestamp] are exchanged
    with ONE fixed-size all-gather per batch; every replica scatters them.

``exchange_touched_rows`` has a pure-torch form for CPU tensors (gloo), which the
world_size-2 CPU tests cover; on the GPU the pack and the scatter are one HIP
kernel each (``zt_pack_rows`` / ``zt_scatter_rows``, tested against the torch
form) and the step never synchronises with the host.
"""
import contextlib
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist


def shard_range(n, rank, world):
    """Contiguous [lo, hi) slice of n items owned by ``rank``."""
    return (n * rank) // world, (n * (rank + 1)) // world


def shard_capacity(n, world):
    return max(shard_range(n, r, world)[1] - shard_range(n, r, world)[0] for r in range(world))


def pack_rows(tables, ids, n_valid, cap):
    """[cap, 1 + sum(widths)] float32 send buffer: id (int32 bits) then each
    table's row.  ids: int32[>=cap]; entries at and beyond n_valid are padding
    and are sent with id = -1."""
    dev = ids.device
    ids = ids[:cap].to(torch.int64)
    valid = torch.arange(cap, device=dev) < n_valid.to(dev).reshape(())
    safe = torch.where(valid, ids, torch.zeros_like(ids))
    cols = [torch.where(valid, ids, torch.full_like(ids, -1)).to(torch.int32).view(torch.float32).reshape(cap, 1)]
    for t in tables:
        rows = t.index_select(0, safe).reshape(cap, -1).to(torch.float32)
        cols.append(torch.where(valid.reshape(cap, 1), rows, torch.zeros_like(rows)))      # padding slots: zeros
    return torch.cat(cols, dim=1).contiguous()


def unpack_rows(tables, recv):
    """Scatter every valid received row (id >= 0) into the local tables."""
    ids = recv[:, 0].contiguous().view(torch.int32).to(torch.int64)
    sel = torch.nonzero(ids >= 0, as_tuple=False).reshape(-1)      # one host sync
    if sel.numel() == 0:
        return 0
    idx = ids.index_select(0, sel)
    col = 1
    for t in tables:
        w = t[0].numel() if t.dim() > 1 else 1
        rows = recv[:, col:col + w].index_select(0, sel)
        if t.dim() == 1:
            t.index_copy_(0, idx, rows.reshape(-1).to(t.dtype))
        else:
            t.index_copy_(0, idx, rows.reshape(-1, *t.shape[1:]).to(t.dtype))
        col += w
    return int(sel.numel())


def _row_tables(tables):
    from ._capi import RowTables
    rt = RowTables()
    rt.n = len(tables)
    width = 1
    for q, t in enumerate(tables):
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise ValueError("exchange tables must be contiguous float32")
        rt.ptr[q] = t.data_ptr()
        rt.width[q] = t[0].numel() if t.dim() > 1 else 1
        width += rt.width[q]
    return rt, width


def pack_rows_device(tables, ids, n_valid, cap):
    """pack_rows on the GPU (one HIP kernel, no host sync)."""
    from ._capi import lib, check, ptr, stream_ptr
    rt, width = _row_tables(tables)
    out = torch.empty((cap, width), dtype=torch.float32, device=ids.device)
    nv = n_valid.reshape(1).to(device=ids.device, dtype=torch.int32)
    check(lib().zt_pack_rows(C.byref(rt), ptr(ids), ptr(nv), C.c_int64(cap), ptr(out), stream_ptr()), "zt_pack_rows")
    return out


def unpack_rows_device(tables, recv):
    """unpack_rows on the GPU (one HIP kernel, no host sync)."""
    from ._capi import lib, check, ptr, stream_ptr
    rt, width = _row_tables(tables)
    assert recv.shape[1] == width and recv.is_contiguous()
    check(lib().zt_scatter_rows(C.byref(rt), ptr(recv), C.c_int64(recv.shape[0]), stream_ptr()), "zt_scatter_rows")


def exchange_touched_rows(tables, ids, n_valid, cap, group=None):
    """All-gather the rows ``ids[:n_valid]`` of every table in ``tables`` from
    all ranks and write them into the local copies.  Fixed-size payload
    (cap rows per rank), one collective.  CUDA tensors: pack and scatter are HIP
    kernels and nothing synchronises with the host (CPU tensors: torch ops)."""
    world = dist.get_world_size(group)
    if ids.is_cuda:
        send = pack_rows_device(tables, ids, n_valid, cap)
        if dist.get_backend(group) == "gloo":
            # rehearsal only (several ranks sharing one GPU in tests): gloo has no GPU all-gather
            recv_h = torch.empty((world * cap, send.shape[1]), dtype=torch.float32)
            dist.all_gather_into_tensor(recv_h, send.cpu(), group=group)
            recv = recv_h.to(send.device)
        else:
            recv = torch.empty((world * cap, send.shape[1]), dtype=torch.float32, device=send.device)
            dist.all_gather_into_tensor(recv, send, group=group)
        unpack_rows_device(tables, recv)
        return recv[:, 0].contiguous().view(torch.int32)       # ids of every row written (-1 = padding)
    send = pack_rows(tables, ids, n_valid, cap)
    recv = torch.empty((world * cap, send.shape[1]), dtype=torch.float32, device=send.device)
    dist.all_gather_into_tensor(recv, send, group=group)
    return unpack_rows(tables, recv)


def allreduce_gradients(params, group=None, bucket_bytes=64 << 20):
    """Data-parallel training: SUM-all-reduce the gradients of ``params`` over the group, in
    flat buckets (the model has ~0.44 M parameters = 1.8 MB: one bucket, one collective; on GPUs the backend is
    "nccl" = RCCL, a ring over xGMI).  Parameters without a gradient on this rank contribute zeros, so every
    rank issues the same collectives.  Returns the number of collectives."""
    params = [p for p in params if p.requires_grad]
    if not params or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return 0
    n_coll, i = 0, 0
    while i < len(params):
        bucket, size = [], 0
        while i < len(params) and (not bucket or size + params[i].numel() * 4 <= bucket_bytes):
            bucket.append(params[i]); size += params[i].numel() * 4; i += 1
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).float() for p in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        o = 0
        for p in bucket:
            g = flat[o:o + p.numel()].reshape(p.shape).to(p.dtype)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            o += p.numel()
        n_coll += 1
    return n_coll


class ShardedTGN:
    """Eval-mode step of a replicated TGN with sharded aggregation / memory update."""

    def __init__(self, tgn, rank, world, group=None):
        self.tgn, self.rank, self.world, self.group = tgn, rank, world, group

    def train_step(self, source_nodes, destination_nodes, negative_nodes, edge_times, edge_idxs, n_neighbors, criterion):
        """One data-parallel training step in the reference's style (train.py:205-213, without the optimizer):
        every rank applies the whole batch to its replica of the state (T-PPR, memory, messages -- no
        communication, replicas stay identical), embeds and scores only its share of the edges, and the
        gradients are summed over the ranks.  The loss of the batch is the mean over ALL its edges, so the
        result equals the single-GPU step.  Returns this rank's share of the loss (sum over ranks = the loss)."""
        tgn = self.tgn
        B = len(source_nodes)
        e0, e1 = shard_range(B, self.rank, self.world)
        sel = np.arange(e0, e1)
        s, d, n = tgn.compute_temporal_embeddings(source_nodes, destination_nodes, negative_nodes, edge_times, edge_idxs,
                                                  n_neighbors, True, edge_sel=sel)
        score = tgn.affinity_score(torch.cat([s, s], dim=0), torch.cat([d, n])).squeeze(dim=0)
        m = e1 - e0
        pos, neg = score[:m].sigmoid(), score[m:].sigmoid()
        dev = pos.device
        # criterion is a mean over its inputs (BCELoss): weight this rank's mean by its share of the batch
        loss = (criterion(pos.squeeze(-1), torch.ones(m, device=dev)) +
                criterion(neg.squeeze(-1), torch.zeros(m, device=dev))) * (m / float(B)) if m else score.sum() * 0.0
        loss.backward()
        allreduce_gradients(tgn.parameters(), self.group)
        return loss.detach()

    @torch.no_grad()
    def step_device(self, src_d, dst_d, neg_d, ts_d, eidx_d, check_status=False, prefetch=None, plan=None, ahead=None):
        tgn = self.tgn
        B = src_d.numel()
        r0, r1 = shard_range(3 * B, self.rank, self.world)
        p0, p1 = shard_range(2 * B, self.rank, self.world)
        main = getattr(tgn, "main_stream", None)
        ctx = torch.cuda.stream(main) if main is not None else contextlib.nullcontext()
        with ctx:
            # P1 streaming: whole batch on every rank (replicas stay bit-identical); P1 pruning, P2: rows [r0, r1);
            # P3: winners at positions [p0, p1) -- only this rank's winners are flagged, the GRU compacts them
            emb = tgn.step_device(src_d, dst_d, neg_d, ts_d, eidx_d, check_status=check_status, prefetch=prefetch,
                                  plan=plan, rows=(r0, r1), positions=(p0, p1), ahead=ahead)
            if getattr(tgn, "_xchg", None) is not None:         # the native step has exchanged the rows itself (enable_exchange)
                return emb
            rows, count = tgn.memory_updater.last_rows()
            m = tgn.memory
            got = exchange_touched_rows([m.memory, m.last_update, m.messages, m.timestamps], rows, count,
                                        shard_capacity(2 * B, self.world), self.group)
            hook = getattr(m, "_rows_changed", None)
            if hook is not None and torch.is_tensor(got):      # rows the other ranks rewrote: the projected table follows
                hook(got, None, got.numel())
        return emb
