# Cost of the wide streaming path (csrc/tppr_wide.hpp: one wavefront per model, edges in order) per edge and model, rows full:
#   python tools/exp/wide_k_cost.py
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from zebra_amd import tppr, synth
for k in (20, 63, 64, 100, 255):
    N, bs, nb = 300, 1000, 8
    src, dst, ts, eidx = synth.power_law_stream(N, bs * nb, seed=5)
    neg = synth.negatives(dst, len(src), seed=6)
    f = tppr.tppr_finder(N + 1, k, 2, [0.1, 0.1], [0.5, 0.95])
    d = torch.device('cuda')
    sd, dd, nd = [torch.from_numpy(x).to(d) for x in (src, dst, neg)]
    td, ed = torch.from_numpy(ts).to(d), torch.from_numpy(eidx).to(d)
    t = []
    for b in range(nb):
        s, e = b * bs, (b + 1) * bs
        torch.cuda.synchronize(); t0 = time.perf_counter()
        f.stream_device(torch.cat([sd[s:e], dd[s:e], nd[s:e]]), td[s:e], ed[s:e], 3, True, -1, check_status=False)
        torch.cuda.synchronize(); t.append(time.perf_counter() - t0)
    f.check_status()
    ln = f.export_state(0)["len"]
    print("k=%3d: %.1f us per edge (both models side by side) over the last 4 batches of %d edges; mean row length %.1f" % (
        k, 1e6 * np.mean(t[-4:]) / bs, bs, ln[ln > 0].mean()), flush=True)
