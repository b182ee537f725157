# host-side profile of the training step of bench.py's c2_train leg (cProfile, top of the cumulative list)
import cProfile, pstats, sys, os, types, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from zebra_amd import synth
wl = dict(synth.WORKLOADS["c2"]); bs = wl["bs"]
n = 200
src, dst, neg, ts, eidx = bench.make_stream(wl, n * bs, perm_seed=7)
dev = torch.device("cuda")
tgn = bench.build_model(wl, dev, n * bs + 1)
tgn.train()
if os.environ.get("ZT_NO_OVERLAY_OP"):
    tgn.embedding_module.overlay_rows_op = False      # A/B: the batch's own rows by torch index / where
opt = torch.optim.Adam(tgn.parameters(), lr=1e-4)
crit = torch.nn.BCELoss()
ones, zeros = torch.ones(bs, device=dev), torch.zeros(bs, device=dev)
def step(b):
    s_, e_ = b * bs, (b + 1) * bs
    opt.zero_grad()
    pos, negp = tgn.compute_edge_probabilities(src[s_:e_], dst[s_:e_], neg[s_:e_], ts[s_:e_], eidx[s_:e_], 10, True)
    loss = crit(pos.squeeze(), ones) + crit(negp.squeeze(), zeros)
    loss.backward()
    opt.step()
    tgn.memory.detach_memory()
for b in range(60): step(b)
torch.cuda.synchronize()
t0 = time.perf_counter()
for b in range(60, 100): step(b)
torch.cuda.synchronize()
print("ms/step %.3f" % (1e3 * (time.perf_counter() - t0) / 40))
pr = cProfile.Profile(); pr.enable()
for b in range(100, 160): step(b)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(45)
