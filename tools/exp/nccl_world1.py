# The multi-rank code path with RCCL itself, as far as one GPU allows: a process group of ONE rank over "nccl",
# ShardedTGN over the native pipeline with CU masks; results must equal the plain run.
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/tests/golden')
import inputs as I
from helpers import build_tgn
from zebra_amd.distributed import ShardedTGN
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29577")
dist.init_process_group("nccl", rank=0, world_size=1)
N, E, D, F, T, k, al, be, seed, bs = 600, 2400, 100, 4, 100, 20, [0.1, 0.1], [0.5, 0.95], 301, 200
src, dst, neg, ts, eidx = I.make_stream("bipartite", N, E, seed)
w = I.model_weights(D, F, T, 2, seed)
_, efeat = I.random_tables(N, E + 1, D, F, seed)
dev = torch.device("cuda")
t = [torch.from_numpy(x).to(dev) for x in (src, dst, neg, ts, eidx)]
batches = [tuple(x[b * bs:(b + 1) * bs] for x in t) for b in range(E // bs)]
outs = []
for mode in ("plain", "sharded_nccl"):
    tgn = build_tgn(N, E + 1, D, F, T, k, al, be, w, efeat).eval()
    embs = []
    if mode == "plain":
        for cur in batches:
            embs.append(tgn.step_device(*cur).cpu().numpy())
    else:
        tgn.enable_pipeline(tppr_cus=64, max_batch=bs, group=2)
        runner = ShardedTGN(tgn, 0, 1)
        with torch.cuda.stream(tgn.main_stream):
            for b, cur in enumerate(batches):
                embs.append(runner.step_device(*cur, ahead=batches[b + 1:b + 6]).cpu().numpy())
        torch.cuda.synchronize()
        tgn.enable_pipeline(False)
    outs.append((embs, tgn.memory.memory.cpu().numpy()))
ok = all(np.array_equal(a, b) for a, b in zip(outs[0][0], outs[1][0])) and np.array_equal(outs[0][1], outs[1][1])
print("nccl world=1 sharded pipeline equals plain:", ok)
dist.destroy_process_group()
sys.exit(0 if ok else 1)
