"""TGN orchestration over the HIP hot path (mirror of model/tgn_model.py:14-232).

``compute_temporal_embeddings`` / ``compute_edge_probabilities`` keep the
reference's signatures (numpy batches in, tensors out) and its exact ordering
of embedding, message store and memory update.  ``step_device`` is the same
eval-mode protocol with device-resident inputs and no host synchronisation:
it is what bench.py times.
"""
import collections
import ctypes as C
import logging

import numpy as np
import torch

from . import _capi
from ._capi import check, lib, ptr, stream_ptr
from .modules import Memory, MergeLayer, TimeEncode, get_embedding_module, get_memory_updater


class TGN(torch.nn.Module):
    def __init__(self, neighbor_finder, node_features, edge_features, device, n_layers=2, n_heads=2, dropout=0.1,
                 use_memory=False, node_dimension=100, time_dimension=100, memory_dimension=100,
                 embedding_module_type="diffusion", message_function="identity", n_neighbors=None,
                 aggregator_type="last", memory_updater_type="gru", use_destination_embedding_in_message=False,
                 use_source_embedding_in_message=False, args=None):
        super().__init__()
        if message_function != "identity":
            raise ValueError("the reference only runs with the identity message function (tgn_model.py:64)")
        if use_destination_embedding_in_message or use_source_embedding_in_message:
            raise ValueError("embedding-in-message variants are not on the accelerated path")
        self.batch_counter = 0
        self.n_layers = n_layers
        self.neighbor_finder = neighbor_finder
        self.device = torch.device(device)
        self.logger = logging.getLogger(__name__)
        self.args = args
        self.test_mode = False

        if torch.is_tensor(edge_features):
            self.edge_raw_features = edge_features.to(self.device, torch.float32).contiguous()
        else:
            self.edge_raw_features = torch.from_numpy(np.asarray(edge_features).astype(np.float32)).to(self.device)
        self.n_edge_features = self.edge_raw_features.shape[1]
        self.n_nodes = args.n_nodes
        self.time_dimension = time_dimension
        self.memory_dimension = memory_dimension
        self.embedding_dimension = node_dimension
        self.n_node_features = self.embedding_dimension
        self.n_neighbors = n_neighbors
        self.embedding_module_type = embedding_module_type
        self.use_destination_embedding_in_message = use_destination_embedding_in_message
        self.use_source_embedding_in_message = use_source_embedding_in_message

        self.time_encoder = TimeEncode(dimension=self.time_dimension)
        self.use_memory = use_memory
        raw_message_dimension = 2 * self.memory_dimension + self.n_edge_features + self.time_dimension
        message_dimension = raw_message_dimension
        self.memory = Memory(n_nodes=self.n_nodes, memory_dimension=self.memory_dimension,
                             input_dimension=message_dimension, message_dimension=message_dimension,
                             device=self.device,
                             reference_compat_aliasing=getattr(args, "reference_compat_aliasing", False))
        self.memory_updater = get_memory_updater(module_type=memory_updater_type, message_dimension=message_dimension,
                                                 memory_dimension=self.memory_dimension, device=self.device)
        self.embedding_module = get_embedding_module(
            module_type=embedding_module_type, node_features=None, edge_features=self.edge_raw_features,
            memory=self.memory, neighbor_finder=self.neighbor_finder, time_encoder=self.time_encoder,
            n_layers=self.n_layers, n_node_features=self.n_node_features, n_edge_features=self.n_edge_features,
            n_time_features=self.time_dimension, embedding_dimension=self.embedding_dimension, device=self.device,
            n_heads=n_heads, dropout=dropout, use_memory=use_memory, n_neighbors=self.n_neighbors, args=args,
            num_nodes=self.n_nodes)
        hidden_dim = self.n_node_features * (len(args.alpha_list) + 1)
        self.affinity_score = MergeLayer(hidden_dim, hidden_dim, hidden_dim, 1)
        # device scratch of the message-store kernel: last occurrence per node (all -1 between calls)
        self._scratch = torch.full((self.n_nodes,), -1, dtype=torch.int32, device=self.device)
        self._status = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.reset_timer()

    def reset_timer(self):
        self.t_get_message = self.t_store_message = self.t_clear_message = self.t_message = 0
        self.t_embedding = self.t_get_memory = self.t_update_memory = self.t_temporal = self.t_score = 0
        self.n_update_memory = 0
        self.embedding_module.t_tppr = 0

    # ------------------------------------------------------------------ device protocol
    def store_messages_device(self, src_d, dst_d, ts_d, eidx_d, pos_range=None):
        """get_raw_messages + store_raw_messages (tgn_model.py:204-226) on the device.
        ``pos_range`` = (lo, hi) restricts the work to winners at batch positions lo..hi-1.
        The touched nodes are exactly those whose flag is set afterwards; the GRU update
        finds them from the (duplicated) endpoint list, no compaction needed here."""
        B = src_d.numel()
        lo, hi = pos_range if pos_range is not None else (0, 2 * B)
        m = self.memory
        check(lib().zt_store_messages_range(ptr(m.memory), ptr(m.last_update), ptr(self.edge_raw_features),
                                      ptr(self.time_encoder.w.weight), C.c_int64(m.n_nodes),
                                      C.c_int64(self.edge_raw_features.shape[0]), C.c_int32(self.memory_dimension),
                                      C.c_int32(self.n_edge_features), C.c_int32(self.time_dimension), ptr(src_d),
                                      ptr(dst_d), ptr(ts_d), ptr(eidx_d), C.c_int64(B), C.c_int64(lo), C.c_int64(hi),
                                      ptr(m.messages),
                                      ptr(m.timestamps), ptr(m._flag_buf), ptr(self._scratch), None,
                                      None, ptr(self._status), stream_ptr()), "zt_store_messages")
        return B

    # ------------------------------------------------------------------ link scorer (tgn_model.py:185-188) on the device
    def _affinity_state(self, max_B, who="direct"):
        """(workspace, weights struct, weights_ready) of the HIP scorer for one CONSUMER -- "direct": score_device, which
        packs the weights in the call that finds them changed; "pipe": the native pipeline, which packs them in its next
        whole-batch step, on its own stream -- ; None when the hidden width has no kernel.  Each consumer has its own
        workspace (packed weights, partial scores, arrival counters) and its own packed-for key: the two pack at different
        times and run on different streams (round-4 advisor)."""
        a = self.affinity_score
        H = a.fc1.weight.shape[0]
        need = lib().zt_affinity_workspace_bytes(C.c_int64(max(1, max_B)), C.c_int32(H))
        if need < 0:
            return None
        allst = getattr(self, "_aff", None)
        if allst is None:
            allst = self._aff = {}
        st = allst.get(who)
        key = tuple((t.data_ptr(), t._version) for t in (a.fc1.weight, a.fc1.bias, a.fc2.weight, a.fc2.bias))
        if st is None or st["ws"].numel() < need or st["max_B"] < max_B:
            st = allst[who] = dict(ws=torch.empty(int(need), dtype=torch.uint8, device=self.device), max_B=int(max_B), key=None)
        w = _capi.AffinityWeights()
        w.fc1_w, w.fc1_b, w.fc2_w, w.fc2_b = (a.fc1.weight.data_ptr(), a.fc1.bias.data_ptr(), a.fc2.weight.data_ptr(),
                                              a.fc2.bias.data_ptr())
        ready = st["key"] == key
        st["key"] = key
        return st, w, ready

    @torch.no_grad()
    def score_device(self, emb):
        """sigmoid(affinity_score([src | src], [dst | neg])) for the [3B, H] embeddings of a batch -> float32 [2B]:
        the B positive probabilities, then the B negative ones (compute_edge_probabilities' tail, eval mode)."""
        B = emb.shape[0] // 3
        st = self._affinity_state(B)
        if st is None:
            score = self.affinity_score(torch.cat([emb[:B], emb[:B]]), emb[B:]).squeeze(1)
            return score.sigmoid()
        st, w, ready = st
        prob = torch.empty(2 * B, dtype=torch.float32, device=self.device)
        check(lib().zt_affinity(ptr(emb.contiguous()), C.c_int64(B), C.c_int32(emb.shape[1]), C.byref(w), ptr(prob), ptr(st["ws"]),
                                C.c_int64(st["max_B"]), C.c_int32(1 if ready else 0), stream_ptr()), "zt_affinity")
        return prob

    def enable_scoring(self, on=True):
        """With the native pipeline: every whole-batch step also scores its 2B pairs behind its aggregation
        (zt_pipeline_set_scoring); ``last_prob()`` returns the last step's probabilities."""
        self._score_on = bool(on)
        self._pipe_scoring_sync(force=True)

    def last_prob(self):
        """float32 [2B] view of the last scored step's probabilities (B positive pairs, then B negative ones), valid on
        the caller's current stream; the buffer is reused two steps later."""
        out, B = C.c_void_p(), C.c_int64()
        check(lib().zt_pipeline_last_scores(self._pipe, stream_ptr(), C.byref(out), C.byref(B)), "zt_pipeline_last_scores")
        off = (out.value - self._prob_buf.data_ptr()) // 4
        return self._prob_buf[off: off + 2 * B.value]

    def _pipe_scoring_sync(self, force=False):
        if getattr(self, "_pipe", None) is None:
            return
        if not getattr(self, "_score_on", False):
            if force:
                check(lib().zt_pipeline_set_scoring(self._pipe, None, None, None), "zt_pipeline_set_scoring")
            return
        max_b = self._pipe_args[1]
        st = self._affinity_state(max_b, "pipe")
        if st is None:
            raise ValueError("no HIP scorer for hidden width %d" % self.affinity_score.fc1.weight.shape[0])
        st, w, ready = st
        if ready and not force and getattr(self, "_pipe_score_set", False):
            return
        if getattr(self, "_prob_buf", None) is None or self._prob_buf.numel() < 4 * max_b:
            self._prob_buf = torch.zeros(4 * max_b, dtype=torch.float32, device=self.device)
        check(lib().zt_pipeline_set_scoring(self._pipe, C.byref(w), ptr(st["ws"]), ptr(self._prob_buf)), "zt_pipeline_set_scoring")
        self._pipe_score_set = True

    # -- the step as ONE native call (csrc/pipeline.hip): P1 of batch b+1 on a side stream beside P2 + P3 of batch b
    def enable_pipeline(self, on=True, tppr_cus=0, max_batch=16384, group=1):
        """Create (or drop) the native step pipeline.  ``tppr_cus`` > 0 pins the T-PPR stream to the first
        tppr_cus compute units (CU mask) and everything else to the rest; callers run their own work on
        ``self.main_stream``.  With it, ``step_device`` takes ``prefetch`` (the NEXT batch: its T-PPR query is
        issued at once -- it must be the batch of the next call) and ``plan`` (the one after it), or ``ahead`` = the
        list of batches that follow, in order.  ``group`` > 1 (streaming strategy): the T-PPR update of that many
        consecutive batches runs as one launch (zt_pipeline_set_group); 3 * group + 1 batches of ``ahead`` keep it full
        (synth.pipeline_look)."""
        if getattr(self, "_pipe", None) is not None:
            torch.cuda.synchronize(self.device)
            check(lib().zt_pipeline_destroy(self._pipe))
        if getattr(self, "_xchg", None) is not None:
            check(lib().zt_exchange_destroy(self._xchg))
        self._xchg = None
        self._xchg_args = None
        self._pipe = None
        self._pipe_sig = None
        self._pipe_keep = None
        self._pipe_stats = None
        self._batch_cache = collections.OrderedDict()
        self.main_stream = None
        if not on:
            return
        self._pipe_args = (int(tppr_cus), int(max_batch))
        self._pipe_group = int(group)
        self._pipe_refresh(create=True)
        self._pipe_score_set = False
        self._pipe_scoring_sync(force=True)

    # -- multi-GPU: the row exchange INSIDE the native step (csrc/exchange.hip); SURVEY.md 8e
    def enable_exchange(self, rank, world, transport="rccl", with_messages=False, group=None, shm_name=None):
        """Every step of the native pipeline ends with the all-gather of the memory rows the ranks rewrote (and their scatter
        and projected-row refresh), enqueued by the library itself: ``step_device(rows=, positions=)`` and ``run_device`` of
        a sharded run then need no torch.distributed call between steps.  transport "rccl": one rank per GPU, the
        communicator's id is broadcast over ``group`` (torch.distributed); "shm": ranks sharing ONE GPU (tests, rehearsals),
        a POSIX shared-memory segment named ``shm_name``.  ``with_messages``: also exchange the message rows (only needed
        to compare that table across ranks: the eval protocol never reads another rank's messages).  Collective."""
        import torch.distributed as dist
        if getattr(self, "_pipe", None) is None:
            raise RuntimeError("enable_exchange needs enable_pipeline()")
        max_b = self._pipe_args[1]
        m = self.memory
        d = _capi.ExchangeDesc()
        d.rank, d.world, d.with_messages = int(rank), int(world), 1 if with_messages else 0
        d.cap_rows = (2 * max_b + world - 1) // world
        keep = None
        if transport == "rccl":
            uid = torch.zeros(128, dtype=torch.uint8)
            if rank == 0:
                check(lib().zt_exchange_unique_id(ptr(uid), C.c_int64(128)), "zt_exchange_unique_id")
            if world > 1:
                on_gpu = dist.get_backend(group) == "nccl"
                t = uid.to(self.device) if on_gpu else uid
                dist.broadcast(t, 0, group=group)
                uid = t.cpu()
            keep = uid.numpy().tobytes()
            d.transport = _capi.XCHG_RCCL
            d.unique_id = C.cast(C.c_char_p(keep), C.c_void_p)
        elif transport == "shm":
            if not shm_name:
                raise ValueError("transport 'shm' needs shm_name")
            keep = shm_name.encode()
            d.transport = _capi.XCHG_SHM
            d.shm_name = keep
        else:
            raise ValueError("transport must be 'rccl' or 'shm'")
        d.memory, d.last_update, d.messages, d.msg_ts = (m.memory.data_ptr(), m.last_update.data_ptr(), m.messages.data_ptr(),
                                                         m.timestamps.data_ptr())
        d.D, d.msg_dim = self.memory_dimension, m.messages.shape[1]
        h = C.c_void_p()
        if transport == "shm" and world > 1 and dist.is_available() and dist.is_initialized():
            # rank 0 makes the segment (an old one of that name is removed first, the new one starts as zeros); the others
            # open it once it exists: never a segment a killed run left behind (csrc/exchange.hip)
            if rank == 0:
                check(lib().zt_exchange_create(C.byref(h), C.byref(d)), "zt_exchange_create")
            dist.barrier(group=group)
            if rank != 0:
                check(lib().zt_exchange_create(C.byref(h), C.byref(d)), "zt_exchange_create")
        else:
            check(lib().zt_exchange_create(C.byref(h), C.byref(d)), "zt_exchange_create")
        self._xchg = h
        self._xchg_args = (int(rank), int(world), bool(with_messages))
        check(lib().zt_pipeline_set_exchange(self._pipe, h), "zt_pipeline_set_exchange")

    def _pipe_signature(self):
        """Everything the native pipeline holds pointers to: (tables, weight versions)."""
        # identity of every table / handle object and in-place version of every weight (the objects are kept
        # alive by _pipe_keep, so an id cannot be recycled while it is part of the signature)
        em, m, g = self.embedding_module, self.memory, self.memory_updater.memory_updater
        tables = (id(m.memory), m.memory._version, id(m.last_update), id(m.messages), id(m.timestamps), id(m._flag_buf),
                  id(self.edge_raw_features), getattr(em, "_proj_serial", 0),
                  id(em._proj["table"]) if em._proj is not None else 0, id(em._status),
                  id(em._ws), id(self.memory_updater._ws),      # the workspaces the descriptor points into (they are
                                                                # replaced when a call outside the pipeline needs more)
                  id(em.tppr_finder._live) if em.tppr_strategy == "streaming" else id(em.neighbor_finder))
        weights = tuple((id(t), t._version) for t in (em.fc1.weight, em.fc1.bias, em.fc2.weight, em.fc2.bias,
                                                      em.fc1_source.weight, em.fc1_source.bias, em.fc2_source.weight,
                                                      em.fc2_source.bias, g.weight_ih, g.weight_hh, g.bias_ih, g.bias_hh))
        return tables, weights

    def _pipe_refresh(self, create=False):
        """(Re)build the descriptor when a table was replaced (restore_memory, __init_memory__, a new neighbour
        finder, restore_tppr) or a weight changed in place; cheap when nothing did.  Runs on the main stream."""
        sig = self._pipe_signature()
        if not create and sig == self._pipe_sig:
            return
        em, m, mu = self.embedding_module, self.memory, self.memory_updater
        tppr_cus, max_b = self._pipe_args
        D, F, T = self.memory_dimension, self.n_edge_features, self.time_dimension
        em._ws_shape = max(3 * max_b, em._ws_shape or 0)
        ws, _, _ = em._workspace(em._ws_shape)
        gws = mu._workspace(2 * max_b, D)
        table = em._projection(m)                       # rebuilt here if the memory or W_m changed
        if em._status is None:
            em._status = torch.zeros(1, dtype=torch.int32, device=self.device)
        d = _capi.PipelineDesc()
        if em.tppr_strategy == "streaming":
            d.tppr = em.tppr_finder._live.h
        else:
            d.csr = em.neighbor_finder._h
            d.width, d.depth = em.width, em.depth
            for q, (a, bb) in enumerate(zip(em.alpha_list, em.beta_list)):
                d.alpha[q], d.beta[q] = float(a), float(bb)
        d.memory, d.last_update, d.messages, d.msg_ts = (m.memory.data_ptr(), m.last_update.data_ptr(),
                                                         m.messages.data_ptr(), m.timestamps.data_ptr())
        d.flags, d.scratch, d.efeat = m._flag_buf.data_ptr(), self._scratch.data_ptr(), self.edge_raw_features.data_ptr()
        d.num_nodes, d.num_edges = m.n_nodes, self.edge_raw_features.shape[0]
        d.D, d.F, d.T, d.M, d.k = D, F, T, em.n_tppr, em.k
        d.ew, d.gw = em._embed_weights(), mu._weights()
        d.embed_ws, d.gru_ws = ws.data_ptr(), gws.data_ptr()
        d.proj_table = table.data_ptr() if table is not None else None
        d.status = em._status.data_ptr()
        d.max_B = max_b
        self._pipe_keep = (ws, gws, table, d, m.memory, m.last_update, m.messages, m.timestamps, m._flag_buf,
                           em.tppr_finder._live if em.tppr_strategy == "streaming" else em.neighbor_finder)
        if create:
            h = C.c_void_p()
            check(lib().zt_pipeline_create(C.byref(h), C.byref(d), C.c_int32(tppr_cus)), "zt_pipeline_create")
            self._pipe = h
            if getattr(self, "_pipe_group", 1) > 1:
                check(lib().zt_pipeline_set_group(h, C.c_int32(self._pipe_group)), "zt_pipeline_set_group")
            self.main_stream = torch.cuda.ExternalStream(lib().zt_pipeline_main_stream(h), device=self.device)
        else:
            check(lib().zt_pipeline_update(self._pipe, C.byref(d), C.c_int32(1)), "zt_pipeline_update")
            if getattr(self, "_xchg", None) is not None:      # the exchange follows the tables
                check(lib().zt_exchange_set_tables(self._xchg, ptr(m.memory), ptr(m.last_update), ptr(m.messages),
                                                   ptr(m.timestamps)), "zt_exchange_set_tables")
        em._ws_key = None                               # the pipeline remakes the padded weights at its next step
        mu._ws_key = None
        self._pipe_sig = self._pipe_signature()

    def _batch_struct(self, batch):
        """zt_batch of five device tensors; remembered by the identity of the edge-id tensor (a batch is
        presented three times: as plan, as prefetch, as the current one)."""
        cache = self._batch_cache
        key = id(batch[4])
        hit = cache.get(key)
        if hit is not None and hit[0][4] is batch[4]:
            return hit[1]
        b = _capi.Batch()
        b.src, b.dst, b.neg, b.ts, b.eidx = [t.data_ptr() for t in batch]
        b.B = batch[0].numel()
        # ALL five tensors stay referenced while a slot may still read them on the plan / T-PPR streams (a batch is
        # staged at most 3 * MAX_GROUP steps before its own step; the oldest entries go first)
        while len(cache) >= 64:
            cache.popitem(last=False)
        cache[key] = (tuple(batch), b)
        return b

    def _pipe_step(self, batch, prefetch, plan, rows=None, positions=None, ahead=None):
        """zt_pipeline_step_ahead; returns the embeddings of ``rows`` (valid on the caller's current stream)."""
        if ahead is None:
            ahead = [b for b in (prefetch, plan if prefetch is not None else None) if b is not None]
        caller = torch.cuda.current_stream(self.device)
        if caller.cuda_stream == self.main_stream.cuda_stream:
            return self._pipe_step_main(batch, ahead, rows, positions)
        self.main_stream.wait_stream(caller)            # inputs produced on the caller's stream
        with torch.cuda.stream(self.main_stream):
            out = self._pipe_step_main(batch, ahead, rows, positions)
        caller.wait_stream(self.main_stream)
        out.record_stream(caller)
        return out

    def _ahead_array(self, ahead):
        """ctypes array of zt_batch for the batches that follow; remembered by the identity of its members."""
        key = tuple(id(b[4]) for b in ahead)
        hit = self._batch_cache.get(key)
        if hit is not None and all(x[4] is b[4] for x, b in zip(hit[0], ahead)):
            return hit[1]
        arr = (_capi.Batch * max(1, len(ahead)))()
        for q, b in enumerate(ahead):
            arr[q] = self._batch_struct(b)
        while len(self._batch_cache) >= 64:
            self._batch_cache.popitem(last=False)
        self._batch_cache[key] = ([tuple(b) for b in ahead], arr)
        return arr

    def _pipe_want_stats(self, on):
        """average_topk from the slot's weights (zt_pipeline_set_stats), only while somebody asks for it"""
        if bool(on) == (self._pipe_stats is not None):
            return
        em = self.embedding_module
        if on:
            self._pipe_stats = torch.zeros(1, dtype=torch.float32, device=self.device)
            check(lib().zt_pipeline_set_stats(self._pipe, ptr(self._pipe_stats)), "zt_pipeline_set_stats")
            em._avg_topk_t = self._pipe_stats
        else:
            check(lib().zt_pipeline_set_stats(self._pipe, None), "zt_pipeline_set_stats")
            self._pipe_stats = None

    def pipeline_outstanding(self):
        """Batches whose T-PPR update ran ahead of their step (0 without a pipeline)."""
        if getattr(self, "_pipe", None) is None:
            return 0
        return int(lib().zt_pipeline_outstanding(self._pipe))

    def _pipe_step_main(self, batch, ahead, rows, positions):
        self._pipe_refresh()
        if getattr(self, "_score_on", False):
            self._pipe_scoring_sync()
        B = batch[0].numel()
        r0, r1 = rows if rows is not None else (0, 3 * B)
        p0, p1 = positions if positions is not None else (0, 2 * B)
        em = self.embedding_module
        out = torch.empty((r1 - r0, self.embedding_dimension * (em.n_tppr + 1)), dtype=torch.float32, device=self.device)
        cur = self._batch_struct(batch)
        arr = self._ahead_array(ahead)
        check(lib().zt_pipeline_step_ahead(self._pipe, C.byref(cur), arr, C.c_int32(len(ahead)), C.c_int64(r0),
                                           C.c_int64(r1), C.c_int64(p0), C.c_int64(p1), ptr(out)), "zt_pipeline_step")
        return out

    def prepare_run(self, batches):
        """The ctypes view of a list of batches (five device tensors each) for ``run_device``: made once, outside a timed
        region.  The tensors must stay alive while a run may still read them."""
        arr = (_capi.Batch * max(1, len(batches)))()
        for q, b in enumerate(batches):
            arr[q].src, arr[q].dst, arr[q].neg, arr[q].ts, arr[q].eidx = [x.data_ptr() for x in b]
            arr[q].B = b[0].numel()
        return arr, len(batches), [tuple(b) for b in batches]

    @torch.no_grad()
    def run_device(self, prepared, out=None, look=None):
        """n consecutive eval-mode steps over whole batches as ONE native call (zt_pipeline_run): the batch loop of
        evaluation/evaluation.py:19-45 without a Python iteration per batch.  ``prepared`` = ``prepare_run(batches)``.
        out: None (one [3 B, H] buffer overwritten by every step -- for callers after the scores or the final state), or a
        [n, 3 B, H] float32 tensor that receives every step's embeddings.  Call from ``self.main_stream``; the streaming
        T-PPR state ends exactly behind the last batch given."""
        if getattr(self, "_pipe", None) is None:
            raise RuntimeError("run_device needs enable_pipeline()")
        if not self.test_mode:
            self.update_memory_in_test(self.memory)
            self.test_mode = True
        arr, n, keep = prepared
        if n == 0:
            return out
        self._pipe_refresh()
        if getattr(self, "_score_on", False):
            self._pipe_scoring_sync()
        self._pipe_want_stats(False)
        H = self.embedding_dimension * (self.embedding_module.n_tppr + 1)
        Bmax = max(int(arr[q].B) for q in range(n))
        rows_max = 3 * Bmax
        if getattr(self, "_xchg_args", None) is not None:          # a sharded run: step b writes this rank's rows only
            rk, wd, _ = self._xchg_args
            rows_max = max((3 * int(arr[q].B) * (rk + 1)) // wd - (3 * int(arr[q].B) * rk) // wd for q in range(n))
        if out is None:
            buf = getattr(self, "_run_scratch", None)
            if buf is None or buf.numel() < max(1, rows_max) * H:
                buf = self._run_scratch = torch.empty(max(1, rows_max) * H, dtype=torch.float32, device=self.device)
            stride = 0
        else:
            if out.shape != (n, rows_max, H) or out.dtype != torch.float32 or not out.is_contiguous():
                raise ValueError("out must be a contiguous float32 [n, rows, H] tensor (rows = 3 B, or this rank's shard of them)")
            buf, stride = out, rows_max * H
        self._run_keep = keep
        check(lib().zt_pipeline_run(self._pipe, arr, C.c_int32(n), C.c_int32(3 * self._pipe_group + 1 if look is None else look),
                                    ptr(buf), C.c_int64(stride)), "zt_pipeline_run")
        return out

    @torch.no_grad()
    def step_device(self, src_d, dst_d, neg_d, ts_d, eidx_d, check_status=False, prefetch=None, plan=None,
                    stats=False, rows=None, positions=None, ahead=None):
        """One eval-mode batch (tgn_model.py:124-174 with train=False), inputs
        int32/int32/int32/float64/int64 CUDA tensors, no host sync unless
        ``check_status``.  Returns the [3B, D*(n_tppr+1)] embeddings (or those of ``rows`` = (lo, hi);
        ``positions`` = (lo, hi) restricts the memory update to winners at those batch positions: the two
        are how a multi-GPU run shards a batch).  With ``enable_pipeline``: ``prefetch`` = the next batch's
        five tensors (queried at once), ``plan`` = the one after it -- or ``ahead`` = the list of the batches that
        follow (see ``enable_pipeline(group=)``); call from ``self.main_stream``."""
        if not self.test_mode:
            self.update_memory_in_test(self.memory)
            self.test_mode = True
        em = self.embedding_module
        B = src_d.numel()
        if getattr(self, "_pipe", None) is not None:
            self._pipe_want_stats(stats and rows is None)
            emb = self._pipe_step((src_d, dst_d, neg_d, ts_d, eidx_d), prefetch, plan, rows, positions, ahead)
        else:
            nodes_d = torch.cat([src_d, dst_d, neg_d])
            ts3 = ts_d if em.tppr_strategy == "streaming" else torch.cat([ts_d, ts_d, ts_d])
            r0, r1 = rows if rows is not None else (0, 3 * B)
            if em.tppr_strategy == "pruning" and rows is not None:       # rows are independent: query the shard only
                on, oe, od, ow = em.pruning_topk_device(nodes_d[r0:r1].contiguous(), ts3[r0:r1].contiguous(),
                                                        check_status=check_status)
            else:
                on, oe, od, ow = em.topk_device(nodes_d, ts3, eidx_d, check_status=check_status)
                if stats:                                # average_topk (modules/embedding_module.py:232-233)
                    em._avg_topk_t = ow[0, : 2 * B].sum(dim=1).mean()
                if rows is not None:
                    on, oe, od, ow = [t[:, r0:r1].contiguous() for t in (on, oe, od, ow)]
            emb = em.embed_device(self.memory.memory, nodes_d[r0:r1].contiguous() if rows is not None else nodes_d,
                                  on, oe, od, ow, check_status=check_status, memory_obj=self.memory)
            self.store_messages_device(src_d, dst_d, ts_d, eidx_d, pos_range=positions)
            self.memory_updater.update_device(self.memory, nodes_d[: 2 * B], 2 * B)      # [src | dst], flagged once each
        if check_status:
            if em.tppr_strategy == "streaming":
                em.tppr_finder.check_status()
            st = int(self._status.item()) or (int(em._status.item()) if em._status is not None else 0)
            if st != 0:
                self._status.zero_()
                if em._status is not None:
                    em._status.zero_()
                if st == _capi.ZT_ERR_TIMEOUT:
                    raise _capi.ZebraError("a kernel of the step gave up a bounded in-kernel wait (status %d): the step's memory "
                                           "update is incomplete" % st)
                raise IndexError("node / edge id out of range (status %d)" % st)
        return emb

    # ------------------------------------------------------------------ reference surface
    def compute_temporal_embeddings(self, source_nodes, destination_nodes, negative_nodes, edge_times, edge_idxs,
                                    n_neighbors, train, edge_sel=None):
        self.batch_counter += 1
        n_samples = len(source_nodes)
        positives = np.concatenate([source_nodes, destination_nodes])
        unique_positives = np.unique(positives)
        self.n_update_memory += len(positives)
        d = self.device
        # the batch goes to the device in ONE asynchronous transfer (a training step used to make eight synchronous ones)
        host = [np.ascontiguousarray(source_nodes, np.int32), np.ascontiguousarray(destination_nodes, np.int32),
                np.ascontiguousarray(edge_times, np.float64), np.ascontiguousarray(edge_idxs, np.int64),
                np.ascontiguousarray(unique_positives, np.int32)]
        if negative_nodes is not None:
            host.append(np.ascontiguousarray(negative_nodes, np.int32))
        dev = _capi.to_device(d, host)
        src_d, dst_d, ts_d, eidx_d, upos_d = dev[:5]

        if not train:
            if negative_nodes is None:
                raise ValueError("the accelerated eval path expects negatives (tgn_model.py:132)")
            neg_d = dev[5]
            node_embedding = self.step_device(src_d, dst_d, neg_d, ts_d, eidx_d, check_status=True, stats=True)
        else:
            if self.pipeline_outstanding():
                raise RuntimeError("the step pipeline has queried %d batches ahead: their T-PPR updates are applied "
                                   "already; step them (or rebuild the pipeline) before a training batch"
                                   % self.pipeline_outstanding())
            self.test_mode = False
            nodes = np.concatenate([source_nodes, destination_nodes, negative_nodes])
            timestamps = np.concatenate([edge_times, edge_times, edge_times])
            # edge_sel (data-parallel training): embed the rows of these edges only; state updates cover the batch
            row_sel = None
            if edge_sel is not None:
                es = torch.as_tensor(np.asarray(edge_sel), device=d).long()
                row_sel = torch.cat([es, es + n_samples, es + 2 * n_samples])
                n_samples = int(es.numel())
            on_device = None
            if negative_nodes is not None:
                on_device = (torch.cat([src_d, dst_d, dev[5]]), ts_d.repeat(3), eidx_d)
            node_embedding = self.embedding_module.compute_embedding_tppr_ensemble(
                memory=self.memory, source_nodes=nodes, timestamps=timestamps, edge_idxs=edge_idxs,
                memory_updater=self.memory_updater, train=True, row_sel=row_sel, on_device=on_device)
            # update memory without gradients, THEN collect raw messages (:155-168)
            with torch.no_grad():
                self.memory_updater.update_device(self.memory, upos_d, upos_d.numel())
            with torch.no_grad():
                self.store_messages_device(src_d, dst_d, ts_d, eidx_d)
        self._last_node_embedding = node_embedding if edge_sel is None else None
        return (node_embedding[:n_samples], node_embedding[n_samples:2 * n_samples], node_embedding[2 * n_samples:])

    def compute_edge_probabilities(self, source_nodes, destination_nodes, negative_nodes, edge_times, edge_idxs,
                                   n_neighbors, train):
        n_samples = len(source_nodes)
        s, dd, n = self.compute_temporal_embeddings(source_nodes, destination_nodes, negative_nodes, edge_times,
                                                    edge_idxs, n_neighbors, train)
        if not train and not torch.is_grad_enabled() and s.is_cuda:
            # eval: the HIP scorer (csrc/scoring.hip) on the [3B, H] block the step returned; same shapes as below
            if getattr(self, "_pipe", None) is not None and getattr(self, "_score_on", False):
                prob = self.last_prob()                                  # the native step has scored this batch already
                return prob[:n_samples].unsqueeze(1), prob[n_samples:].unsqueeze(1)
            full = getattr(self, "_last_node_embedding", None)           # the [3B, H] block s / dd / n are slices of
            if full is None or full.data_ptr() != s.data_ptr() or full.shape[0] != 3 * n_samples:
                full = torch.cat([s, dd, n])
            prob = self.score_device(full)
            return prob[:n_samples].unsqueeze(1), prob[n_samples:].unsqueeze(1)
        score = self.affinity_score(torch.cat([s, s], dim=0), torch.cat([dd, n])).squeeze(dim=0)
        return score[:n_samples].sigmoid(), score[n_samples:].sigmoid()

    def update_memory(self, memory, positives):
        with torch.no_grad():
            self.memory_updater.update_memory(memory, positives)

    def update_memory_in_test(self, memory):
        with torch.no_grad():
            self.memory_updater.update_memory_in_test(memory)

    def get_updated_memory(self, memory):
        return self.memory_updater.get_updated_memory(memory)

    def set_neighbor_finder(self, neighbor_finder):
        self.neighbor_finder = neighbor_finder
        self.embedding_module.neighbor_finder = neighbor_finder
