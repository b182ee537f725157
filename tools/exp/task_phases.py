# Where does a GENERAL task of k_stream spend its time?  (-DZT_STAMP build, tools/build_stamp.sh: per task of model 0 the
# 100 MHz clock at dequeue, rows ready, first row stored, end.)  T-PPR alone, C5's stream, warm state.
#   python tools/exp/task_phases.py [batches]
import sys, ctypes as C
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from zebra_amd import _capi
_capi.LIB_PATH = '/root/repo/tools/out/libzebra_stamp.so'
from zebra_amd import tppr, synth
lib = _capi.lib()
wl = synth.WORKLOADS["c5"]; B = 4096; NB = int(sys.argv[1]) if len(sys.argv) > 1 else 120
src, dst, ts, eidx = synth.power_law_stream(wl["n_nodes"], NB * B, seed=2020)
neg = synth.negatives(dst, len(src), seed=2021)
f = tppr.tppr_finder(wl["n_nodes"] + 1, 20, 2, [0.1, 0.1], [0.5, 0.95])
d = torch.device('cuda')
sd, dd, nd = [torch.from_numpy(x).to(d) for x in (src, dst, neg)]
td, ed = torch.from_numpy(ts).to(d), torch.from_numpy(eidx).to(d)
for b in range(NB):
    s, e = b * B, (b + 1) * B
    f.stream_device(torch.cat([sd[s:e], dd[s:e], nd[s:e]]), td[s:e], ed[s:e], 3, True, -1, check_status=False)
f.check_status()
st = np.zeros((B, 4), np.int64)
lib.zt_debug_stamps(st.ctypes.data_as(C.c_void_p), C.c_int(B))
s0 = (NB - 1) * B
u, v = src[s0:s0 + B], dst[s0:s0 + B]
cnt = np.bincount(np.concatenate([u, v]))
hot = np.argsort(cnt)[-16:]                                   # the chains' nodes: their edges are not general tasks
gen = ~(np.isin(u, hot) | np.isin(v, hot)) & (st[:, 3] > 0) & (st[:, 0] > 0)
t = st[gen].astype(np.float64) * 0.01                        # us
p = lambda a: np.percentile(a, [10, 50, 90]).round(1)
print("%d general tasks of the last batch: dequeue -> rows ready %s us ; -> first row stored %s ; -> end %s ; whole task %s (mean %.1f)" % (
    gen.sum(), p(t[:, 1] - t[:, 0]), p(t[:, 2] - t[:, 1]), p(t[:, 3] - t[:, 2]), p(t[:, 3] - t[:, 0]), (t[:, 3] - t[:, 0]).mean()))
