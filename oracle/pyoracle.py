"""ctypes front-end of oracle/libzebra_oracle.so.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Never imported by zebra_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libzebra_oracle.so")
_lib = None


def build(force=False):
    """Compile the oracle with gcc (recipe: oracle/Makefile)."""
    if force or not os.path.exists(_LIB_PATH):
        subprocess.run(["make", "-C", _HERE] + (["-B"] if force else []), check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.zo_tppr_create.restype = C.c_void_p
        _lib.zo_numba_int_pow.restype = C.c_double
        _lib.zo_find_before.restype = C.c_int64
        _lib.zo_store_messages.restype = C.c_int64
        _lib.zo_gru_update.restype = C.c_int64
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def numba_argsort(values):
    v = _c(values, np.float64)
    r = np.empty(len(v), np.int32)
    lib().zo_numba_argsort(_p(v), C.c_int32(len(v)), _p(r))
    return r


def numba_int_pow(a, b):
    return lib().zo_numba_int_pow(C.c_double(a), C.c_int64(b))


class TpprOracle:
    """Mirror of the reference's tppr_finder (utils/util.py:391) on the C oracle."""

    def __init__(self, num_nodes, k, n_tppr, alpha_list, beta_list):
        self.num_nodes, self.k, self.n_tppr = int(num_nodes), int(k), int(n_tppr)
        self._a = _c(alpha_list, np.float64)
        self._b = _c(beta_list, np.float64)
        self._h = C.c_void_p(lib().zo_tppr_create(C.c_int64(num_nodes), C.c_int32(k), C.c_int32(n_tppr),
                                                  _p(self._a), _p(self._b)))
        if not self._h:
            raise MemoryError("zo_tppr_create failed")

    def __del__(self):
        if getattr(self, "_h", None):
            lib().zo_tppr_destroy(self._h)
            self._h = None

    def reset_tppr(self):
        lib().zo_tppr_reset(self._h)

    def clone(self):
        o = TpprOracle(self.num_nodes, self.k, self.n_tppr, self._a, self._b)
        lib().zo_tppr_copy(o._h, self._h)
        return o

    def copy_from(self, other):
        if lib().zo_tppr_copy(self._h, other._h) != 0:
            raise ValueError("shape mismatch")

    def _stream(self, nodes, ts, eidx, n_roles, emit, model):
        nodes = _c(nodes, np.int32)
        ts = _c(ts, np.float64)
        eidx = _c(eidx, np.int64)
        B = len(nodes) // n_roles
        nm = self.n_tppr if model < 0 else 1
        rows = n_roles * B
        if emit:
            on = np.empty((nm, rows, self.k), np.int32)
            oe = np.empty((nm, rows, self.k), np.int32)
            od = np.empty((nm, rows, self.k), np.float32)
            ow = np.empty((nm, rows, self.k), np.float32)
        else:
            on = oe = od = ow = None
        rc = lib().zo_tppr_stream(self._h, _p(nodes), _p(ts), _p(eidx), C.c_int64(B), C.c_int32(n_roles),
                                  C.c_int32(1 if emit else 0), C.c_int32(model), _p(on), _p(oe), _p(od), _p(ow))
        if rc != 0:
            raise IndexError("node id out of range (rc=%d)" % rc)
        return on, oe, od, ow

    def streaming_topk(self, source_nodes, timestamps, edge_idxs):
        on, oe, od, ow = self._stream(source_nodes, timestamps, edge_idxs, 3, True, -1)
        return list(on), list(oe), list(od), list(ow)

    def streaming_topk_no_fake(self, source_nodes, timestamps, edge_idxs):
        on, oe, od, ow = self._stream(source_nodes, timestamps, edge_idxs, 2, True, -1)
        return list(on), list(oe), list(od), list(ow)

    def single_streaming_topk(self, source_nodes, timestamps, edge_idxs, tppr_id):
        on, oe, od, ow = self._stream(source_nodes, timestamps, edge_idxs, 3, True, int(tppr_id))
        return on[0], oe[0], od[0], ow[0]

    def streaming_topk_threads(self, source_nodes, timestamps, edge_idxs):
        """The same with one host thread per T-PPR model (the models are independent; inside a model the
        reference's loop is sequential).  bench.py's N-thread P1 variant; N = n_tppr."""
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(self.n_tppr) as ex:
            res = list(ex.map(lambda m: self._stream(source_nodes, timestamps, edge_idxs, 3, True, m),
                              range(self.n_tppr)))
        return tuple([r[q][0] for r in res] for q in range(4))

    def import_rows(self, m, ids, st):
        ids = _c(ids, np.int64)
        rc = lib().zo_tppr_import_rows(self._h, C.c_int32(m), _p(ids), C.c_int64(len(ids)), _p(_c(st["len"], np.int32)),
                                       _p(_c(st["norm"], np.float64)), _p(_c(st["eidx"], np.int64)),
                                       _p(_c(st["node"], np.int64)), _p(_c(st["ts"], np.float64)),
                                       _p(_c(st["w"], np.float64)))
        if rc != 0:
            raise IndexError("zo_tppr_import_rows rc=%d" % rc)

    def export_rows(self, m, ids):
        ids = _c(ids, np.int64)
        n, k = len(ids), self.k
        ln = np.empty(n, np.int32)
        norm = np.empty(n, np.float64)
        e = np.empty((n, k), np.int64)
        nd = np.empty((n, k), np.int64)
        ts = np.empty((n, k), np.float64)
        w = np.empty((n, k), np.float64)
        rc = lib().zo_tppr_export_rows(self._h, C.c_int32(m), _p(ids), C.c_int64(n), _p(ln), _p(norm), _p(e), _p(nd),
                                       _p(ts), _p(w))
        if rc != 0:
            raise IndexError("zo_tppr_export_rows rc=%d" % rc)
        return dict(len=ln, norm=norm, eidx=e, node=nd, ts=ts, w=w)

    def update_only(self, sources, targets, timestamps, edge_idxs):
        nodes = np.concatenate([_c(sources, np.int32), _c(targets, np.int32)])
        self._stream(nodes, timestamps, edge_idxs, 2, False, -1)

    def export(self, m):
        N, k = self.num_nodes, self.k
        ln = np.empty(N, np.int32)
        norm = np.empty(N, np.float64)
        e = np.empty((N, k), np.int64)
        nd = np.empty((N, k), np.int64)
        ts = np.empty((N, k), np.float64)
        w = np.empty((N, k), np.float64)
        lib().zo_tppr_export(self._h, C.c_int32(m), _p(ln), _p(norm), _p(e), _p(nd), _p(ts), _p(w))
        return dict(len=ln, norm=norm, eidx=e, node=nd, ts=ts, w=w)


class CsrOracle:
    """get_neighbor_finder + NeighborFinder (utils/util.py:90-276) on the C oracle."""

    def __init__(self, sources, destinations, edge_idxs, timestamps, num_nodes=None):
        src = _c(sources, np.int32)
        dst = _c(destinations, np.int32)
        eidx = _c(edge_idxs, np.int64)
        ts = _c(timestamps, np.float64)
        E = len(src)
        if num_nodes is None:
            num_nodes = int(max(src.max(), dst.max())) + 1 if E else 1
        self.num_nodes = int(num_nodes)
        self.indptr = np.empty(self.num_nodes + 1, np.int64)
        self.nbr = np.empty(2 * E, np.int32)
        self.eid = np.empty(2 * E, np.int32)
        self.ts = np.empty(2 * E, np.float64)
        rc = lib().zo_csr_build(_p(src), _p(dst), _p(eidx), _p(ts), C.c_int64(E), C.c_int64(self.num_nodes),
                                _p(self.indptr), _p(self.nbr), _p(self.eid), _p(self.ts))
        if rc != 0:
            raise IndexError("node id out of range")

    def find_before(self, v, t):
        n = lib().zo_find_before(_p(self.indptr), _p(self.ts), C.c_int32(int(v)), C.c_double(float(t)))
        lo = self.indptr[v]
        return self.nbr[lo:lo + n], self.eid[lo:lo + n], self.ts[lo:lo + n]

    def get_pruned_topk(self, source_nodes, timestamps, width, depth, alpha, beta, k,
                        node_list, edge_idxs_list, delta_time_list, weight_list):
        q = _c(source_nodes, np.int32)
        t = _c(timestamps, np.float64)
        for a, dt in ((node_list, np.int32), (edge_idxs_list, np.int32), (delta_time_list, np.float32),
                      (weight_list, np.float32)):
            assert a.dtype == dt and a.flags.c_contiguous and a.shape == (len(q), k)
        rc = lib().zo_pruned_topk(_p(self.indptr), _p(self.nbr), _p(self.eid), _p(self.ts),
                                  C.c_int64(self.num_nodes), _p(q), _p(t), C.c_int64(len(q)),
                                  C.c_int32(width), C.c_int32(depth), C.c_double(alpha), C.c_double(beta),
                                  C.c_int32(k), _p(node_list), _p(edge_idxs_list), _p(delta_time_list),
                                  _p(weight_list))
        if rc != 0:
            raise IndexError("get_pruned_topk failed rc=%d" % rc)


def embed(memory, efeat, time_w, nodes, nbr, eix, dt, w, weights, n_threads=1):
    """zo_embed; nbr/eix/dt/w are [M][N][k]; weights = dict of torch-layout arrays."""
    memory = _c(memory, np.float32)
    efeat = _c(efeat, np.float32)
    time_w = _c(time_w, np.float32)
    nodes = _c(nodes, np.int32)
    nbr = _c(nbr, np.int32)
    eix = _c(eix, np.int32)
    dt = _c(dt, np.float32)
    w = _c(w, np.float32)
    M, N, k = nbr.shape
    D = memory.shape[1]
    F = efeat.shape[1]
    T = len(time_w)
    ws = {n: _c(weights[n], np.float32) for n in
          ("fc1_w", "fc1_b", "fc2_w", "fc2_b", "fc1s_w", "fc1s_b", "fc2s_w", "fc2s_b")}
    assert ws["fc1_w"].shape == (D, D + F + T)
    out = np.empty((N, D * (M + 1)), np.float32)
    rc = lib().zo_embed(_p(memory), _p(efeat), _p(time_w), C.c_int64(memory.shape[0]), C.c_int64(efeat.shape[0]),
                        C.c_int32(D), C.c_int32(F), C.c_int32(T), _p(nodes), C.c_int64(N), C.c_int32(M),
                        C.c_int32(k), _p(nbr), _p(eix), _p(dt), _p(w), _p(ws["fc1_w"]), _p(ws["fc1_b"]),
                        _p(ws["fc2_w"]), _p(ws["fc2_b"]), _p(ws["fc1s_w"]), _p(ws["fc1s_b"]), _p(ws["fc2s_w"]),
                        _p(ws["fc2s_b"]), _p(out), C.c_int32(n_threads))
    if rc != 0:
        raise IndexError("zo_embed: id out of range")
    return out


class MemoryOracle:
    """Memory + GRU updater state (modules/memory.py, modules/memory_updater.py)."""

    def __init__(self, n_nodes, D, msg_dim):
        self.n_nodes, self.D, self.msg_dim = n_nodes, D, msg_dim
        self.memory = np.zeros((n_nodes, D), np.float32)
        self.last_update = np.zeros(n_nodes, np.float32)
        self.messages = np.zeros((n_nodes, msg_dim), np.float32)
        self.timestamps = np.zeros(n_nodes, np.float32)
        self.flags = np.zeros(n_nodes, np.uint8)
        self._scratch = np.full(n_nodes, -1, np.int32)

    def store_messages(self, efeat, time_w, src, dst, ts, eidx):
        efeat = _c(efeat, np.float32)
        time_w = _c(time_w, np.float32)
        src = _c(src, np.int32)
        dst = _c(dst, np.int32)
        ts = _c(ts, np.float64)
        eidx = _c(eidx, np.int64)
        F, T = efeat.shape[1], len(time_w)
        assert 2 * self.D + F + T == self.msg_dim
        rc = lib().zo_store_messages(_p(self.memory), _p(self.last_update), _p(efeat), _p(time_w),
                                     C.c_int64(self.n_nodes), C.c_int64(efeat.shape[0]), C.c_int32(self.D),
                                     C.c_int32(F), C.c_int32(T), _p(src), _p(dst), _p(ts), _p(eidx),
                                     C.c_int64(len(src)), _p(self.messages), _p(self.timestamps), _p(self.flags),
                                     _p(self._scratch))
        if rc < 0:
            raise IndexError("zo_store_messages: id out of range")
        return rc

    def gru_update(self, gru, ids=None, n_threads=1):
        ws = {n: _c(gru[n], np.float32) for n in ("w_ih", "w_hh", "b_ih", "b_hh")}
        if ids is not None:
            ids = _c(ids, np.int32)
        rc = lib().zo_gru_update(_p(self.memory), _p(self.last_update), _p(self.messages), _p(self.timestamps),
                                 _p(self.flags), C.c_int64(self.n_nodes), C.c_int32(self.D), C.c_int32(self.msg_dim),
                                 _p(ids), C.c_int64(0 if ids is None else len(ids)), _p(ws["w_ih"]), _p(ws["w_hh"]),
                                 _p(ws["b_ih"]), _p(ws["b_hh"]), C.c_int32(n_threads))
        if rc < 0:
            raise IndexError("zo_gru_update: id out of range")
        return rc


def affinity(x1, x2, w):
    x1 = _c(x1, np.float32)
    x2 = _c(x2, np.float32)
    ws = {n: _c(w[n], np.float32) for n in ("fc1_w", "fc1_b", "fc2_w", "fc2_b")}
    out = np.empty(x1.shape[0], np.float32)
    lib().zo_affinity(_p(x1), _p(x2), C.c_int64(x1.shape[0]), C.c_int32(x1.shape[1]), _p(ws["fc1_w"]),
                      _p(ws["fc1_b"]), _p(ws["fc2_w"]), _p(ws["fc2_b"]), _p(out))
    return out


class ProtocolOracle:
    """The per-batch protocol of TGN.compute_temporal_embeddings / compute_edge_probabilities
    (model/tgn_model.py:124-188) composed from the oracle's pieces, eval AND train mode
    (train: lazily updated memory for the selected neighbours, modules/embedding_module.py:227-230,
    modules/memory_updater.py:61-90; eager update + clear BEFORE the messages, tgn_model.py:155-157).
    ``finder`` is a TpprOracle (streaming) or a CsrOracle (pruning; swap with set_neighbor_finder)."""

    def __init__(self, n_nodes, D, F, T, k, alpha, beta, weights, efeat, time_w, strategy="streaming", finder=None,
                 width=10, depth=2, n_threads=1):
        self.N, self.D, self.F, self.T, self.k = n_nodes, D, F, T, k
        self.alpha, self.beta = list(alpha), list(beta)
        self.w, self.efeat, self.tw = weights, _c(efeat, np.float32), _c(time_w, np.float32)
        self.strategy, self.width, self.depth, self.n_threads = strategy, width, depth, n_threads
        self.gru = {kk: weights[kk] for kk in ("w_ih", "w_hh", "b_ih", "b_hh")}
        self.aff = dict(fc1_w=weights["aff1_w"], fc1_b=weights["aff1_b"], fc2_w=weights["aff2_w"],
                        fc2_b=weights["aff2_b"])
        self.tppr = TpprOracle(n_nodes, k, len(alpha), alpha, beta) if strategy == "streaming" else None
        self.finder = finder
        self.init_memory()
        self.test_mode = False
        self.p23 = "c"                            # "torch": P2 / P3 through oracle/torch_cpu.py (same state arrays)
        self._t23 = None

    def init_memory(self):                       # Memory.__init_memory__ (modules/memory.py:19-25)
        self.mem = MemoryOracle(self.N, self.D, 2 * self.D + self.F + self.T)

    def set_neighbor_finder(self, finder):       # tgn_model.py:230-232
        self.finder = finder

    def topk(self, nodes, ts, eidx):
        if self.strategy == "streaming":
            return self.tppr.streaming_topk(nodes, ts, eidx)
        n = len(nodes)
        t3 = np.concatenate([ts] * (n // len(ts)))
        outs = ([], [], [], [])
        for a, b in zip(self.alpha, self.beta):   # modules/embedding_module.py:280-297
            arr = (np.zeros((n, self.k), np.int32), np.zeros((n, self.k), np.int32),
                   np.zeros((n, self.k), np.float32), np.zeros((n, self.k), np.float32))
            self.finder.get_pruned_topk(nodes, t3, self.width, self.depth, a, b, self.k, *arr)
            for o, x in zip(outs, arr):
                o.append(x)
        return outs

    def _torch_batch(self, src, dst, neg, ts, eidx, topk_out):
        """The eval-mode batch with P2 / P3 on torch-CPU ops (bench.py's cpu_baseline)."""
        if self._t23 is None or self._t23.mem is not self.mem:
            from torch_cpu import TorchCpuP23
            self._t23 = TorchCpuP23(self.mem, self.w, self.efeat, self.tw, self.n_threads)
        t, B = self._t23, len(src)
        nodes = np.concatenate([src, dst, neg]).astype(np.int32)
        positives = np.unique(np.concatenate([src, dst]))
        if not self.test_mode:
            t.gru_update(None)
            self.test_mode = True
        on, oe, od, ow = topk_out if topk_out is not None else self.topk(nodes, ts, eidx)
        emb = t.embed(nodes, on, oe, od, ow)
        t.store_messages(src, dst, ts, eidx)
        t.gru_update(positives)
        self.average_topk = float(np.mean(np.sum(ow[0][:2 * B], axis=1)))
        return emb, None

    def batch(self, src, dst, neg, ts, eidx, train, topk_out=None):
        """-> (embeddings [3B, D*(M+1)], probabilities [2B]).  ``topk_out``: this batch's T-PPR query if the
        caller has run (and timed) it already."""
        if self.p23 == "torch" and not train:
            return self._torch_batch(src, dst, neg, ts, eidx, topk_out)
        mem, B = self.mem, len(src)
        nodes = np.concatenate([src, dst, neg]).astype(np.int32)
        positives = np.unique(np.concatenate([src, dst]))
        if train:
            self.test_mode = False
        elif not self.test_mode:
            mem.gru_update(self.gru, None, n_threads=self.n_threads)          # update_memory_in_test (:142-146)
            self.test_mode = True
        on, oe, od, ow = topk_out if topk_out is not None else self.topk(nodes, ts, eidx)
        table = mem.memory
        if train:                                                            # get_updated_memory(memory, index)
            index = np.unique(np.concatenate([a.ravel() for a in on]))
            lazy = MemoryOracle(self.N, self.D, mem.msg_dim)
            lazy.memory, lazy.last_update = mem.memory.copy(), mem.last_update.copy()
            lazy.messages, lazy.timestamps, lazy.flags = mem.messages, mem.timestamps, mem.flags.copy()
            lazy.gru_update(self.gru, index, n_threads=self.n_threads)
            table = lazy.memory
        emb = embed(table, self.efeat, self.tw, nodes, np.stack(on), np.stack(oe), np.stack(od), np.stack(ow),
                    self.w, n_threads=self.n_threads)
        if train:
            mem.gru_update(self.gru, positives, n_threads=self.n_threads)     # :155-157
        mem.store_messages(self.efeat, self.tw, src, dst, ts, eidx)           # :159-168
        if not train:
            mem.gru_update(self.gru, positives, n_threads=self.n_threads)     # :170-172
        prob = affinity(np.concatenate([emb[:B], emb[:B]]), np.concatenate([emb[B:2 * B], emb[2 * B:]]), self.aff)
        self.average_topk = float(np.mean(np.sum(ow[0][:2 * B], axis=1)))
        return emb, prob

    def backup_memory(self):                     # modules/memory.py:49-50 (flags aliased, as in the reference)
        m = self.mem
        return m.memory.copy(), m.last_update.copy(), m.messages.copy(), m.flags, m.timestamps.copy()

    def restore_memory(self, b):                 # modules/memory.py:52-53
        m = self.mem
        m.memory, m.last_update, m.messages, m.flags, m.timestamps = b[0].copy(), b[1].copy(), b[2].copy(), b[3], \
            b[4].copy()
