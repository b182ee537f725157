"""torch-CPU restatement of P2 (gather + TimeEncode + transform + weighted sum) and P3 (last-message store, GRU
memory update) -- the "build's own torch-CPU module" SURVEY.md 8(d) prescribes for the timed CPU baseline.

TEST INFRASTRUCTURE ONLY (like everything under oracle/): imported by tests/ and by bench.py's cpu_baseline
leg, never by zebra_amd/.  It runs the same torch ops the reference runs (index-select gathers, F.linear = BLAS
GEMM, cos, GRUCell), on the host's cores, IN PLACE on a pyoracle.MemoryOracle's numpy arrays (torch.from_numpy
shares them), so that it can take turns with the C port on one state.  Checked against the C port in
tests/test_oracle_golden.py.

Reference lines: modules/embedding_module.py:243-276,320-328 (embed), model/time_encoding.py:18-28,
model/tgn_model.py:204-226 + modules/memory.py:27-30 (messages), modules/memory_updater.py:29-57,95-98 (GRU).
"""
import numpy as np
import torch
import torch.nn.functional as F


class TorchCpuP23:
    def __init__(self, mem, weights, efeat, time_w, n_threads=None):
        """mem: pyoracle.MemoryOracle (state shared, updated in place); weights: dict of torch-layout arrays."""
        if n_threads:
            torch.set_num_threads(int(n_threads))
        self.mem = mem
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32))
        self.w = {k: t(v) for k, v in weights.items()}
        self.efeat = t(efeat)
        self.tw = t(time_w).view(-1)
        self.D = mem.D

    def _views(self):
        m = self.mem          # (the oracle may have replaced its arrays: take fresh views every call)
        return (torch.from_numpy(m.memory), torch.from_numpy(m.last_update), torch.from_numpy(m.messages),
                torch.from_numpy(m.timestamps), torch.from_numpy(m.flags))

    @torch.no_grad()
    def embed(self, nodes, on, oe, od, ow):
        """compute_embedding_tppr_ensemble, eval mode: nodes int[n]; on/oe/od/ow: per model [n][k]."""
        memory = self._views()[0]
        w = self.w
        idx = torch.from_numpy(np.asarray(nodes).astype(np.int64))
        src = F.linear(F.relu(F.linear(memory[idx], w["fc1s_w"], w["fc1s_b"])), w["fc2s_w"], w["fc2s_b"])   # :243-246
        outs = [src]
        for m in range(len(on)):
            nbr = memory[torch.from_numpy(np.asarray(on[m]).astype(np.int64))]                    # [n, k, D]
            ef = self.efeat[torch.from_numpy(np.asarray(oe[m]).astype(np.int64))]                  # [n, k, F]
            te = torch.cos(torch.from_numpy(np.ascontiguousarray(od[m], np.float32)).unsqueeze(-1) * self.tw)   # [n, k, T]
            x = torch.cat([nbr, ef, te], dim=2)                                                   # :264
            x = F.linear(F.relu(F.linear(x, w["fc1_w"], w["fc1_b"])), w["fc2_w"], w["fc2_b"])      # :320-322, eval
            wt = torch.from_numpy(np.ascontiguousarray(ow[m], np.float32))
            s = wt.sum(dim=1, keepdim=True)
            wn = torch.where(s != 0, wt / torch.where(s != 0, s, torch.ones_like(s)), torch.zeros_like(wt))   # :268-272
            outs.append((x * wn.unsqueeze(-1)).sum(dim=1))
        return torch.cat(outs, dim=1).numpy()

    @torch.no_grad()
    def store_messages(self, src, dst, ts, eidx):
        """get_raw_messages + store_raw_messages: the LAST occurrence of a node in [src | dst] wins."""
        memory, last_update, messages, timestamps, flags = self._views()
        B = len(src)
        nodes = np.concatenate([src, dst]).astype(np.int64)
        partner = np.concatenate([dst, src]).astype(np.int64)
        uniq, first_rev = np.unique(nodes[::-1], return_index=True)            # :208-211
        pos = 2 * B - 1 - first_rev
        ids = torch.from_numpy(uniq)
        par = torch.from_numpy(partner[pos])
        e = torch.from_numpy(np.asarray(eidx).astype(np.int64)[pos % B])
        tf = torch.from_numpy(np.asarray(ts, np.float64)[pos % B]).float()     # edge_times .float() (:213)
        delta = tf - last_update[ids]                                          # :221
        te = torch.cos(delta.unsqueeze(-1) * self.tw)
        messages[ids] = torch.cat([memory[ids], memory[par], self.efeat[e], te], dim=1)
        timestamps[ids] = tf
        flags[ids] = 1
        return len(uniq)

    @torch.no_grad()
    def gru_update(self, ids=None):
        """update_memory (ids) / update_memory_in_test (None) + clear_messages."""
        memory, last_update, messages, timestamps, flags = self._views()
        if ids is None:
            sel = torch.nonzero(flags).view(-1)
            flags.zero_()
        else:
            idx = torch.from_numpy(np.asarray(ids).astype(np.int64))
            sel = idx[flags[idx] != 0]
            flags[idx] = 0
        if sel.numel() == 0:
            return 0
        w = self.w
        h = torch._VF.gru_cell(messages[sel], memory[sel], w["w_ih"], w["w_hh"], w["b_ih"], w["b_hh"])   # nn.GRUCell
        memory[sel] = h
        last_update[sel] = timestamps[sel]
        return int(sel.numel())
