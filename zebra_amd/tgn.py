"""TGN orchestration over the HIP hot path (mirror of model/tgn_model.py:14-232).

``compute_temporal_embeddings`` / ``compute_edge_probabilities`` keep the
reference's signatures (numpy batches in, tensors out) and its exact ordering
of embedding, message store and memory update.  ``step_device`` is the same
eval-mode protocol with device-resident inputs and no host synchronisation:
it is what bench.py times.
"""
import ctypes as C
import logging

import numpy as np
import torch

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

    # -- P1 on a side stream: the T-PPR state depends only on the edge stream, never on the node
    # -- memory, so the query of batch b+1 can run while batch b is being aggregated / written back.
    def enable_pipeline(self, on=True, tppr_cus=0):
        """Run the T-PPR query on its own HIP stream and allow ``prefetch`` in step_device.
        ``tppr_cus`` > 0 pins that stream to the first tppr_cus compute units (CU mask) and the
        caller should run everything else on ``self.main_stream`` (the remaining CUs)."""
        self._pending = None
        self._planned = {}
        self._plan_stream = None
        self.main_stream = None
        for hptr in getattr(self, "_masked", []):          # CU-masked streams of an earlier call
            torch.cuda.synchronize(self.device)
            check(lib().zt_stream_destroy(C.c_void_p(hptr)))
        self._masked = []
        if not on:
            self._side = None
            return
        self._plan_stream = torch.cuda.Stream(device=self.device)
        if tppr_cus > 0:
            n_cu = torch.cuda.get_device_properties(self.device).multi_processor_count
            hs, hm = C.c_void_p(), C.c_void_p()
            check(lib().zt_stream_create_masked(C.byref(hs), C.c_int32(0), C.c_int32(tppr_cus)))
            check(lib().zt_stream_create_masked(C.byref(hm), C.c_int32(tppr_cus), C.c_int32(n_cu)))
            self._masked = [hs.value, hm.value]
            self._side = torch.cuda.ExternalStream(hs.value, device=self.device)
            self.main_stream = torch.cuda.ExternalStream(hm.value, device=self.device)
        else:
            self._side = torch.cuda.Stream(device=self.device)

    def plan_batch(self, batch):
        """Pipeline mode, streaming T-PPR: the dependency prepass of a batch still two steps away, on
        its own stream (it reads only the ids and overlaps the update kernel of the batch before)."""
        em = self.embedding_module
        if getattr(self, "_side", None) is None or em.tppr_strategy != "streaming":
            return
        src_d, dst_d, neg_d, _, eidx_d = batch
        with torch.cuda.stream(self._plan_stream):
            nodes_d = torch.cat([src_d, dst_d, neg_d])
            token = em.tppr_finder.plan_device(nodes_d, eidx_d, 3, -1)
        nodes_d.record_stream(self._side)
        self._planned[(eidx_d.data_ptr(), eidx_d.numel())] = (nodes_d, token)

    def _tppr_launch(self, src_d, dst_d, neg_d, ts_d, eidx_d, check_status):
        em = self.embedding_module
        nodes_d, token = getattr(self, "_planned", {}).pop((eidx_d.data_ptr(), eidx_d.numel()), (None, 0))
        if nodes_d is None:
            nodes_d = torch.cat([src_d, dst_d, neg_d])
        ts3 = ts_d if em.tppr_strategy == "streaming" else torch.cat([ts_d, ts_d, ts_d])
        return (nodes_d,) + tuple(em.topk_device(nodes_d, ts3, eidx_d, check_status=check_status, plan_token=token))

    def tppr_batch_device(self, batch, prefetch=None, check_status=False, plan=None):
        """T-PPR query of ``batch`` = (src, dst, neg, ts, eidx) -> (nodes, nbr, eidx, dt, w) device
        tensors, valid on the current stream.  With the pipeline enabled, ``prefetch`` (the NEXT
        batch) is enqueued on the side stream before returning."""
        side = getattr(self, "_side", None)
        if side is None:
            return self._tppr_launch(*batch, check_status)
        main = torch.cuda.current_stream()
        key = (batch[4].data_ptr(), batch[4].numel())
        if self._pending is not None and self._pending[0] == key:
            _, outs, ev = self._pending
        else:
            with torch.cuda.stream(side):
                outs = self._tppr_launch(*batch, False)
                ev = torch.cuda.Event()
                ev.record(side)
        self._pending = None
        main.wait_event(ev)
        for t in outs:
            t.record_stream(main)
        if prefetch is not None:
            with torch.cuda.stream(side):
                nxt = self._tppr_launch(*prefetch, False)
                ev2 = torch.cuda.Event()
                ev2.record(side)
            self._pending = ((prefetch[4].data_ptr(), prefetch[4].numel()), nxt, ev2)
        if plan is not None:
            self.plan_batch(plan)
        if check_status and self.embedding_module.tppr_strategy == "streaming":
            self.embedding_module.tppr_finder.check_status()
        return outs

    @torch.no_grad()
    def step_device(self, src_d, dst_d, neg_d, ts_d, eidx_d, check_status=False, prefetch=None, plan=None,
                    stats=False):
        """One eval-mode batch (tgn_model.py:124-174 with train=False), inputs
        int32/int32/int32/float64/int64 CUDA tensors, no host sync unless
        ``check_status``.  Returns the [3B, D*(n_tppr+1)] embeddings.
        ``prefetch`` = the next batch's five tensors, ``plan`` = the one after it (pipeline mode only)."""
        if not self.test_mode:
            self.update_memory_in_test(self.memory)
            self.test_mode = True
        em = self.embedding_module
        nodes_d, on, oe, od, ow = self.tppr_batch_device((src_d, dst_d, neg_d, ts_d, eidx_d), prefetch, check_status,
                                                        plan)
        if stats:                                    # average_topk (modules/embedding_module.py:232-233)
            em._avg_topk_t = ow[0, : 2 * src_d.numel()].sum(dim=1).mean()
        emb = em.embed_device(self.memory.memory, nodes_d, on, oe, od, ow, check_status=check_status,
                              memory_obj=self.memory)
        B = self.store_messages_device(src_d, dst_d, ts_d, eidx_d)
        self.memory_updater.update_device(self.memory, nodes_d[: 2 * B], 2 * B)      # [src | dst], flagged once each
        if check_status:
            st = int(self._status.item())
            if st != 0:
                self._status.zero_()
                raise IndexError("node / edge id out of range (status %d)" % st)
        return emb

    # ------------------------------------------------------------------ reference surface
    def compute_temporal_embeddings(self, source_nodes, destination_nodes, negative_nodes, edge_times, edge_idxs,
                                    n_neighbors, train):
        self.batch_counter += 1
        n_samples = len(source_nodes)
        positives = np.concatenate([source_nodes, destination_nodes])
        unique_positives = np.unique(positives)
        self.n_update_memory += len(positives)
        d = self.device
        src_d = torch.as_tensor(np.ascontiguousarray(source_nodes, np.int32), device=d)
        dst_d = torch.as_tensor(np.ascontiguousarray(destination_nodes, np.int32), device=d)
        ts_d = torch.as_tensor(np.ascontiguousarray(edge_times, np.float64), device=d)
        eidx_d = torch.as_tensor(np.ascontiguousarray(edge_idxs, np.int64), device=d)

        if not train:
            if negative_nodes is None:
                raise ValueError("the accelerated eval path expects negatives (tgn_model.py:132)")
            neg_d = torch.as_tensor(np.ascontiguousarray(negative_nodes, np.int32), device=d)
            node_embedding = self.step_device(src_d, dst_d, neg_d, ts_d, eidx_d, check_status=True, stats=True)
        else:
            self.test_mode = False
            nodes = np.concatenate([source_nodes, destination_nodes, negative_nodes])
            timestamps = np.concatenate([edge_times, edge_times, edge_times])
            node_embedding = self.embedding_module.compute_embedding_tppr_ensemble(
                memory=self.memory, source_nodes=nodes, timestamps=timestamps, edge_idxs=edge_idxs,
                memory_updater=self.memory_updater, train=True)
            # update memory without gradients, THEN collect raw messages (:155-168)
            self.update_memory(self.memory, unique_positives)
            with torch.no_grad():
                self.store_messages_device(src_d, dst_d, ts_d, eidx_d)
        return (node_embedding[:n_samples], node_embedding[n_samples:2 * n_samples], node_embedding[2 * n_samples:])

    def compute_edge_probabilities(self, source_nodes, destination_nodes, negative_nodes, edge_times, edge_idxs,
                                   n_neighbors, train):
        n_samples = len(source_nodes)
        s, dd, n = self.compute_temporal_embeddings(source_nodes, destination_nodes, negative_nodes, edge_times,
                                                    edge_idxs, n_neighbors, train)
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
