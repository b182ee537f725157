"""Host-side mirror of the reference's T-PPR classes over the HIP library.

Same names, argument order and return conventions as the reference
(utils/util.py): ``tppr_finder`` (:391-873), ``NeighborFinder`` (:144-276) and
``get_neighbor_finder`` (:90-107).  The arithmetic runs in
libzebra_amd.so on the GPU; torch is only used for device buffers and the
current HIP stream.  There is no CPU fallback.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _capi
from ._capi import check, lib, ptr, stream_ptr


def _dev():
    if not torch.cuda.is_available():
        raise RuntimeError("zebra_amd needs a ROCm GPU (MI355X / gfx950); no CPU fallback exists")
    return torch.device("cuda", torch.cuda.current_device())


class _TpprState:
    """Owns one device-resident T-PPR state (a zt_tppr handle)."""

    def __init__(self, num_nodes, k, n_tppr, alpha, beta):
        self.shape = (int(num_nodes), int(k), int(n_tppr))
        self._alpha = np.ascontiguousarray(alpha, np.float64)
        self._beta = np.ascontiguousarray(beta, np.float64)
        h = C.c_void_p()
        check(lib().zt_tppr_create(C.byref(h), C.c_int64(num_nodes), C.c_int32(k), C.c_int32(n_tppr),
                                   ptr(self._alpha), ptr(self._beta)), "zt_tppr_create")
        self.h = h
        self._pid = os.getpid()

    def __del__(self):
        h, self.h = getattr(self, "h", None), None
        # a fork()ed child inherits the Python object but not the GPU context: the handle is the parent's to free
        if h and os is not None and getattr(self, "_pid", None) == os.getpid():     # (os is None at interpreter exit)
            try:
                lib().zt_tppr_destroy(h)
            except Exception:
                pass

    def set_device_share(self, n):
        check(lib().zt_tppr_set_device_share(self.h, C.c_int32(n)), "zt_tppr_set_device_share")
        self._share = int(n)

    def clone(self):
        o = _TpprState(*self.shape, self._alpha, self._beta)
        if getattr(self, "_share", 1) > 1:
            o.set_device_share(self._share)
        check(lib().zt_tppr_copy(o.h, self.h, stream_ptr()), "zt_tppr_copy")
        return o

    def export(self, m):
        N, k, _ = self.shape
        ln = np.empty(N, np.int32)
        norm = np.empty(N, np.float64)
        e = np.empty((N, k), np.int64)
        nd = np.empty((N, k), np.int64)
        ts = np.empty((N, k), np.float64)
        w = np.empty((N, k), np.float64)
        check(lib().zt_tppr_export(self.h, C.c_int32(m), ptr(ln), ptr(norm), ptr(e), ptr(nd), ptr(ts), ptr(w)),
              "zt_tppr_export")
        return dict(len=ln, norm=norm, eidx=e, node=nd, ts=ts, w=w)

    def export_rows(self, m, ids):
        ids = np.ascontiguousarray(ids, np.int64)
        n, k = len(ids), self.shape[1]
        ln = np.empty(n, np.int32)
        norm = np.empty(n, np.float64)
        e = np.empty((n, k), np.int64)
        nd = np.empty((n, k), np.int64)
        ts = np.empty((n, k), np.float64)
        w = np.empty((n, k), np.float64)
        check(lib().zt_tppr_export_rows(self.h, C.c_int32(m), ptr(ids), C.c_int64(n), ptr(ln), ptr(norm), ptr(e),
                                        ptr(nd), ptr(ts), ptr(w)), "zt_tppr_export_rows")
        return dict(len=ln, norm=norm, eidx=e, node=nd, ts=ts, w=w)

    def import_(self, m, st):
        check(lib().zt_tppr_import(self.h, C.c_int32(m), ptr(np.ascontiguousarray(st["len"], np.int32)),
                                   ptr(np.ascontiguousarray(st["norm"], np.float64)),
                                   ptr(np.ascontiguousarray(st["eidx"], np.int64)),
                                   ptr(np.ascontiguousarray(st["node"], np.int64)),
                                   ptr(np.ascontiguousarray(st["ts"], np.float64)),
                                   ptr(np.ascontiguousarray(st["w"], np.float64))), "zt_tppr_import")


    def import_rows(self, m, ids, st):
        ids = np.ascontiguousarray(ids, np.int64)
        check(lib().zt_tppr_import_rows(self.h, C.c_int32(m), ptr(ids), C.c_int64(len(ids)),
                                        ptr(np.ascontiguousarray(st["len"], np.int32)),
                                        ptr(np.ascontiguousarray(st["norm"], np.float64)),
                                        ptr(np.ascontiguousarray(st["eidx"], np.int64)),
                                        ptr(np.ascontiguousarray(st["node"], np.int64)),
                                        ptr(np.ascontiguousarray(st["ts"], np.float64)),
                                        ptr(np.ascontiguousarray(st["w"], np.float64))), "zt_tppr_import_rows")


class tppr_finder:
    """Drop-in for the reference's ``tppr_finder`` jitclass (utils/util.py:391).

    ``reference_compat_aliasing=True`` reproduces the reference's shallow
    backup / restore / val snapshots (they alias the live state);
    the default takes true device-side snapshots.
    """

    def __init__(self, num_nodes, k, n_tppr, alpha_list, beta_list, reference_compat_aliasing=False):
        if len(alpha_list) != n_tppr or len(beta_list) != n_tppr:
            raise ValueError("alpha_list / beta_list must have n_tppr entries")
        self.num_nodes = int(num_nodes)
        self.k = int(k)
        self.n_tppr = int(n_tppr)
        self.alpha_list = [float(a) for a in alpha_list]
        self.beta_list = [float(b) for b in beta_list]
        self.reference_compat_aliasing = bool(reference_compat_aliasing)
        self._dev = _dev()
        self._val = None                     # reset_val_tppr (:399): empty; allocated on first use
        self._live = self._new_state()       # reset_tppr (:400)

    # ------------------------------------------------------------ state mgmt
    def _new_state(self):
        st = _TpprState(self.num_nodes, self.k, self.n_tppr, self.alpha_list, self.beta_list)
        if getattr(self, "_share", 1) > 1:
            st.set_device_share(self._share)
        return st

    def chain_stats(self):
        """Hub-chain statistics of the live state since the last call: dict(pairs_claimed, pairs_done, pairs_left_in_prep,
        pairs_left_in_section, singles) -- how often two chain positions shared one critical section (csrc/tppr_pair.hpp)."""
        out = (C.c_int64 * 5)()
        check(lib().zt_tppr_chain_stats(self._live.h, out, stream_ptr()), "zt_tppr_chain_stats")
        return dict(zip(("pairs_claimed", "pairs_done", "pairs_left_in_prep", "pairs_left_in_section", "singles"), [int(x) for x in out]))

    def set_device_share(self, n_processes):
        """Several processes run their T-PPR updates on ONE GPU (ranks rehearsing on a one-GPU box): every launch takes
        1 / n of its stream's compute units and runs without hub chains (zt_tppr_set_device_share).  One rank per GPU
        -- the supported multi-GPU layout -- needs no call."""
        self._share = int(n_processes)
        self._live.set_device_share(self._share)
        if self._val is not None:
            self._val.set_device_share(self._share)

    def reset_val_tppr(self):                # utils/util.py:402-417
        self._val = None

    def _val_state(self):
        if self._val is None:
            self._val = self._new_state()
        return self._val

    def reset_tppr(self):                    # utils/util.py:419-434
        if self.reference_compat_aliasing:
            self._live = self._new_state()   # new objects; old ones may live on in val/backups
        else:
            check(lib().zt_tppr_reset(self._live.h, stream_ptr()), "zt_tppr_reset")

    def backup_tppr(self):                   # utils/util.py:436-437
        return self._live if self.reference_compat_aliasing else self._live.clone()

    def restore_tppr(self, backup):          # utils/util.py:439-440
        self._live = backup if self.reference_compat_aliasing else backup.clone()

    def restore_val_tppr(self):              # utils/util.py:442-444
        self._live = self._val_state() if self.reference_compat_aliasing else self._val_state().clone()

    # reference attributes, materialised on demand from the device state
    @property
    def norm_list(self):
        return [self._live.export(m)["norm"] for m in range(self.n_tppr)]

    @property
    def PPR_list(self):
        return [self._dicts(self._live, m) for m in range(self.n_tppr)]

    @property
    def val_norm_list(self):
        return [self._val_state().export(m)["norm"] for m in range(self.n_tppr)]

    @property
    def val_PPR_list(self):
        return [self._dicts(self._val_state(), m) for m in range(self.n_tppr)]

    @staticmethod
    def _dicts(state, m):
        st = state.export(m)
        out = []
        for v in range(len(st["len"])):
            n = st["len"][v]
            out.append({(int(st["eidx"][v, j]), int(st["node"][v, j]), float(st["ts"][v, j])): float(st["w"][v, j])
                        for j in range(n)})
        return out

    def export_state(self, m):
        """Dense view of model m's dictionaries (iteration order)."""
        return self._live.export(m)

    def export_rows(self, m, node_ids):
        """The same for the given nodes only: arrays [n], [n][k] (large graphs: the touched nodes)."""
        return self._live.export_rows(m, node_ids)

    # ------------------------------------------------------------- checkpoint
    def state_dict(self, node_ids):
        """The T-PPR state of ``node_ids`` (the nodes the stream has touched so far) for every model, as
        plain numpy arrays: what the reference leaves out of its checkpoint (train.py:291), so that a
        resumed run continues bit for bit instead of replaying the stream (fill_tppr)."""
        ids = np.ascontiguousarray(node_ids, np.int64)
        out = {"node_ids": ids, "k": np.int32(self.k), "n_tppr": np.int32(self.n_tppr)}
        for m in range(self.n_tppr):
            for key, val in self._live.export_rows(m, ids).items():
                out["m%d_%s" % (m, key)] = val
        return out

    def load_state_dict(self, sd, reset=True):
        """Inverse of ``state_dict`` (``reset``: every other node's dictionary is emptied first)."""
        if int(sd["k"]) != self.k or int(sd["n_tppr"]) != self.n_tppr:
            raise ValueError("checkpoint was written for k=%d, n_tppr=%d" % (int(sd["k"]), int(sd["n_tppr"])))
        if reset:
            self.reset_tppr()
        ids = np.ascontiguousarray(sd["node_ids"], np.int64)
        for m in range(self.n_tppr):
            self._live.import_rows(m, ids, {key: sd["m%d_%s" % (m, key)] for key in ("len", "norm", "eidx", "node", "ts", "w")})

    # ------------------------------------------------------------- streaming
    def _upload(self, source_nodes, timestamps, edge_idxs, n_roles):
        nodes = np.ascontiguousarray(source_nodes, np.int32)
        if nodes.ndim != 1 or len(nodes) % n_roles:
            raise ValueError("source_nodes must be 1-D with a multiple of %d entries" % n_roles)
        B = len(nodes) // n_roles
        ts = np.ascontiguousarray(np.asarray(timestamps, np.float64)[:B])
        eidx = np.ascontiguousarray(edge_idxs, np.int64)
        if len(ts) < B or len(eidx) != B:
            raise ValueError("timestamps / edge_idxs shorter than the batch")
        d = self._dev
        return (torch.from_numpy(nodes).to(d), torch.from_numpy(ts).to(d), torch.from_numpy(eidx).to(d), B)

    def stream_device(self, nodes_d, ts_d, eidx_d, n_roles=3, emit=True, model=-1, check_status=True, plan_token=0):
        """Device-resident form: int32[n_roles*B], float64[B], int64[B] CUDA
        tensors in, four [n_models][n_roles*B][k] CUDA tensors out.  ``plan_token``: what
        ``plan_device`` returned for this very call (same ``nodes_d``), or 0."""
        B = nodes_d.numel() // n_roles
        nm = self.n_tppr if model < 0 else 1
        d = nodes_d.device
        if emit:
            on = torch.empty((nm, n_roles * B, self.k), dtype=torch.int32, device=d)
            oe = torch.empty_like(on)
            od = torch.empty((nm, n_roles * B, self.k), dtype=torch.float32, device=d)
            ow = torch.empty_like(od)
        else:
            on = oe = od = ow = None
        rc = lib().zt_tppr_stream(self._live.h, ptr(nodes_d), ptr(ts_d), ptr(eidx_d), C.c_int64(B),
                                  C.c_int32(n_roles), C.c_int32(1 if emit else 0), C.c_int32(model), ptr(on),
                                  ptr(oe), ptr(od), ptr(ow), C.c_uint64(plan_token), stream_ptr())
        self._raise_latched(rc, "zt_tppr_stream")
        if check_status:
            check(lib().zt_tppr_status(self._live.h, stream_ptr()), "zt_tppr_stream")
        return on, oe, od, ow

    def plan_device(self, nodes_d, eidx_d, n_roles=3, model=-1):
        """Runs the dependency prepass of a coming ``stream_device`` call (same ``nodes_d`` tensor,
        same batch) on the CURRENT stream; it reads only the ids, so it can overlap the previous
        call's update kernel.  Returns the plan's token for ``stream_device(plan_token=...)``
        (0: nothing was planned).  Optional: a call without a token runs its own prepass."""
        B = nodes_d.numel() // n_roles
        tok = C.c_uint64(0)
        rc = lib().zt_tppr_plan(self._live.h, ptr(nodes_d), ptr(eidx_d), C.c_int64(B), C.c_int32(n_roles),
                                C.c_int32(model), C.byref(tok), stream_ptr())
        self._raise_latched(rc, "zt_tppr_plan")
        return int(tok.value)

    def _raise_latched(self, rc, what):
        """A failure latched by an EARLIER launch (bad id in a batch run with check_status=False, or a
        dependency time-out) surfaces at the next call: report it once (zt_tppr_status also clears it)."""
        if rc in (_capi.ZT_ERR_RANGE, _capi.ZT_ERR_TIMEOUT):
            check(lib().zt_tppr_status(self._live.h, stream_ptr()) or rc, what)
        check(rc, what)

    def check_status(self):
        check(lib().zt_tppr_status(self._live.h, stream_ptr()), "zt_tppr_stream")

    @staticmethod
    def _lists(on, oe, od, ow):
        return ([a for a in on.cpu().numpy()], [a for a in oe.cpu().numpy()], [a for a in od.cpu().numpy()],
                [a for a in ow.cpu().numpy()])

    def streaming_topk(self, source_nodes, timestamps, edge_idxs):            # utils/util.py:473-576
        n, t, e, _ = self._upload(source_nodes, timestamps, edge_idxs, 3)
        return self._lists(*self.stream_device(n, t, e, 3, True, -1))

    def streaming_topk_no_fake(self, source_nodes, timestamps, edge_idxs):    # utils/util.py:682-782
        n, t, e, _ = self._upload(source_nodes, timestamps, edge_idxs, 2)
        return self._lists(*self.stream_device(n, t, e, 2, True, -1))

    def single_streaming_topk(self, source_nodes, timestamps, edge_idxs, tppr_id):   # utils/util.py:581-679
        n, t, e, _ = self._upload(source_nodes, timestamps, edge_idxs, 3)
        on, oe, od, ow = self.stream_device(n, t, e, 3, True, int(tppr_id))
        return on[0].cpu().numpy(), oe[0].cpu().numpy(), od[0].cpu().numpy(), ow[0].cpu().numpy()

    def compute_val_tppr(self, sources, targets, timestamps, edge_idxs, chunk=1 << 16):   # utils/util.py:787-873
        """One pass over a whole stream without emitting rows, then snapshot
        the result as the ``val`` state."""
        src = np.ascontiguousarray(sources, np.int32)
        dst = np.ascontiguousarray(targets, np.int32)
        ts = np.ascontiguousarray(timestamps, np.float64)
        eidx = np.ascontiguousarray(edge_idxs, np.int64)
        for s in range(0, len(src), chunk):
            e = min(len(src), s + chunk)
            nodes = np.concatenate([src[s:e], dst[s:e]])
            n, t, ei, _ = self._upload(nodes, ts[s:e], eidx[s:e], 2)
            self.stream_device(n, t, ei, 2, False, -1, check_status=False)
        self.check_status()
        self._val = self._live if self.reference_compat_aliasing else self._live.clone()


class NeighborFinder:
    """Drop-in for the reference's ``NeighborFinder`` jitclass (utils/util.py:144)."""

    def __init__(self, node_to_neighbors, node_to_edge_idxs, node_to_edge_timestamps, _handle=None):
        self._dev = _dev()
        self._pid = os.getpid()
        if _handle is not None:
            self._h = _handle
        else:
            n = len(node_to_neighbors)
            indptr = np.zeros(n + 1, np.int64)
            indptr[1:] = np.cumsum([len(a) for a in node_to_neighbors])
            cat = (lambda xs, dt: np.ascontiguousarray(np.concatenate([np.asarray(x, dt) for x in xs]) if n else
                                                       np.zeros(0, dt), dt))
            nbr, eid, ts = cat(node_to_neighbors, np.int32), cat(node_to_edge_idxs, np.int32), \
                cat(node_to_edge_timestamps, np.float64)
            h = C.c_void_p()
            check(lib().zt_csr_from_sorted(C.byref(h), ptr(indptr), ptr(nbr), ptr(eid), ptr(ts), C.c_int64(n)),
                  "zt_csr_from_sorted")
            self._h = h
        n, e2 = C.c_int64(), C.c_int64()
        check(lib().zt_csr_size(self._h, C.byref(n), C.byref(e2)))
        self.num_nodes, self._e2 = n.value, e2.value
        self._indptr = np.empty(self.num_nodes + 1, np.int64)
        self._nbr = np.empty(self._e2, np.int32)
        self._eid = np.empty(self._e2, np.int32)
        self._ts = np.empty(self._e2, np.float64)
        check(lib().zt_csr_export(self._h, ptr(self._indptr), ptr(self._nbr), ptr(self._eid), ptr(self._ts)))
        self._status = torch.zeros(1, dtype=torch.int32, device=self._dev)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and os is not None and getattr(self, "_pid", None) == os.getpid():      # never from a fork()ed child
            try:
                lib().zt_csr_destroy(h)
            except Exception:
                pass

    # reference attributes (lists of per-node arrays, utils/util.py:147-149)
    def _split(self, a):
        return [a[self._indptr[v]:self._indptr[v + 1]] for v in range(self.num_nodes)]

    @property
    def node_to_neighbors(self):
        return self._split(self._nbr)

    @property
    def node_to_edge_idxs(self):
        return self._split(self._eid)

    @property
    def node_to_edge_timestamps(self):
        return self._split(self._ts)

    def find_before(self, src_idx, cut_time):                                 # utils/util.py:152-154
        if src_idx < 0 or src_idx >= self.num_nodes:
            raise IndexError("node id %d out of range" % src_idx)
        lo, hi = self._indptr[src_idx], self._indptr[src_idx + 1]
        i = int(np.searchsorted(self._ts[lo:hi], cut_time))
        return self._nbr[lo:lo + i], self._eid[lo:lo + i], self._ts[lo:lo + i]

    def pruned_topk_device(self, nodes_d, ts_d, width, depth, alpha, beta, k, on, oe, od, ow, check_status=True):
        check(lib().zt_pruned_topk(self._h, ptr(nodes_d), ptr(ts_d), C.c_int64(nodes_d.numel()), C.c_int32(width),
                                   C.c_int32(depth), C.c_double(alpha), C.c_double(beta), C.c_int32(k), ptr(on),
                                   ptr(oe), ptr(od), ptr(ow), ptr(self._status), stream_ptr()), "zt_pruned_topk")
        if check_status:
            st = int(self._status.item())
            if st != 0:
                self._status.zero_()
                raise IndexError("get_pruned_topk: node id out of range (status %d)" % st)

    def pruned_topk_multi_device(self, nodes_d, ts_d, width, depth, alphas, betas, k, on, oe, od, ow, check_status=True):
        """All (alpha, beta) models of the ensemble in one walk (zt_pruned_topk_multi); on/oe/od/ow are [M, n, k]."""
        M = len(alphas)
        a = (C.c_double * M)(*[float(x) for x in alphas])
        b = (C.c_double * M)(*[float(x) for x in betas])
        check(lib().zt_pruned_topk_multi(self._h, ptr(nodes_d), ptr(ts_d), C.c_int64(nodes_d.numel()), C.c_int32(width),
                                         C.c_int32(depth), C.c_int32(M), a, b, C.c_int32(k), ptr(on), ptr(oe), ptr(od),
                                         ptr(ow), ptr(self._status), stream_ptr()), "zt_pruned_topk_multi")
        if check_status:
            st = int(self._status.item())
            if st != 0:
                self._status.zero_()
                raise IndexError("get_pruned_topk: node id out of range (status %d)" % st)

    def get_pruned_topk(self, source_nodes, timestamps, width, depth, alpha, beta, k, node_list, edge_idxs_list,
                        delta_time_list, weight_list):                        # utils/util.py:185-276
        """Writes into the four caller-owned [N, k] arrays in place, returns None."""
        q = torch.from_numpy(np.ascontiguousarray(source_nodes, np.int32)).to(self._dev)
        t = torch.from_numpy(np.ascontiguousarray(timestamps, np.float64)).to(self._dev)
        outs = []
        for a, dt in ((node_list, np.int32), (edge_idxs_list, np.int32), (delta_time_list, np.float32),
                      (weight_list, np.float32)):
            if a.dtype != dt or a.shape != (len(q), k):
                raise ValueError("output arrays must be [N, k] int32/int32/float32/float32")
            outs.append(torch.from_numpy(np.ascontiguousarray(a)).to(self._dev))
        self.pruned_topk_device(q, t, width, depth, float(alpha), float(beta), k, *outs)
        for a, o in zip((node_list, edge_idxs_list, delta_time_list, weight_list), outs):
            a[...] = o.cpu().numpy()


def get_neighbor_finder(data):
    """utils/util.py:90-107: undirected, time-sorted adjacency of a ``Data``
    object (attributes sources, destinations, edge_idxs, timestamps)."""
    _dev()
    src = np.ascontiguousarray(data.sources, np.int32)
    dst = np.ascontiguousarray(data.destinations, np.int32)
    eidx = np.ascontiguousarray(data.edge_idxs, np.int64)
    ts = np.ascontiguousarray(data.timestamps, np.float64)
    n = int(max(src.max(), dst.max())) + 1 if len(src) else 1
    h = C.c_void_p()
    check(lib().zt_csr_build(C.byref(h), ptr(src), ptr(dst), ptr(eidx), ptr(ts), C.c_int64(len(src)), C.c_int64(n)),
          "zt_csr_build")
    return NeighborFinder(None, None, None, _handle=h)
