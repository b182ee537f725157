# C5's T-PPR update alone (product library), NB batches after a short warm-up: for counter collection on k_stream
import sys, ctypes as C
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from zebra_amd import _capi, tppr, synth
lib = _capi.lib()
wl = synth.WORKLOADS["c5"]; B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096; NB = int(sys.argv[1]) if len(sys.argv) > 1 else 60
src, dst, ts, eidx = synth.power_law_stream(wl["n_nodes"], NB * B, seed=2020)
neg = synth.negatives(dst, len(src), seed=2021)
f = tppr.tppr_finder(wl["n_nodes"] + 1, 20, 2, [0.1, 0.1], [0.5, 0.95])
d = torch.device('cuda')
sd, dd, nd = [torch.from_numpy(x).to(d) for x in (src, dst, neg)]
td, ed = torch.from_numpy(ts).to(d), torch.from_numpy(eidx).to(d)
lib.zt_profile_reset(); lib.zt_profile_enable(1)
for b in range(NB):
    s, e = b * B, (b + 1) * B
    f.stream_device(torch.cat([sd[s:e], dd[s:e], nd[s:e]]), td[s:e], ed[s:e], 3, True, -1, check_status=False)
f.check_status()
n, ms = C.c_int64(), C.c_double(); lib.zt_profile_read(b"tppr_stream", C.byref(n), C.byref(ms))
print("B=%d: avg k_stream us: %.1f  per 4096 edges: %.1f" % (B, 1e3 * ms.value / n.value, 1e3 * ms.value / n.value * 4096 / B))
