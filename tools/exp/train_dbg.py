# one configuration of tests/soak_train.py with per-parameter statistics:   python tools/exp/train_dbg.py <seed>
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import soak_train as S
import torch
import inputs as I
from helpers import build_tgn
seed = int(sys.argv[1])
rng = np.random.RandomState(seed)
D, T = [(100, 100), (100, 100), (32, 16), (64, 100)][rng.randint(4)]
F = int(rng.choice([1, 4, 16, 172])); k = int(rng.choice([5, 10, 20, 40])); M = int(rng.choice([1, 2]))
N = int(rng.choice([60, 900, 5000])); bs = int(rng.choice([20, 200, 600])); nb = int(rng.randint(2, 5))
kind = ["bipartite", "general", "hub"][rng.randint(3)]
al = [float(rng.choice([0.1, 0.2])) for _ in range(M)]; be = [float(rng.choice([0.5, 0.8, 0.95])) for _ in range(M)]
E = bs * nb
src, dst, neg, ts, eidx = I.make_stream(kind, N, E, seed)
w = I.model_weights(D, F, T, M, seed); _, efeat = I.random_tables(N, E + 1, D, F, seed)
dev = torch.device("cuda")
G = [torch.from_numpy(np.random.RandomState(seed * 7 + b).standard_normal((3 * bs, (M + 1) * D)).astype(np.float32)).to(dev) for b in range(nb)]
res = {}
for fused in (True, False, "double"):
    tgn = build_tgn(N, E + 1, D, F, T, k, al, be, w, efeat)
    tgn.embedding_module.fused_training = (fused is True)
    tgn.train(True)
    out = []
    for b in range(nb):
        s, e = b * bs, (b + 1) * bs
        tgn.zero_grad()
        se, de, ne = tgn.compute_temporal_embeddings(src[s:e], dst[s:e], neg[s:e], ts[s:e], eidx[s:e], 10, True)
        emb = torch.cat([se, de, ne])
        (emb * G[b]).sum().backward()
        out.append((emb.detach().cpu().numpy(), {pn: p.grad.detach().cpu().numpy().copy() for pn, p in tgn.named_parameters() if p.grad is not None}))
    res[fused] = out
for b in range(nb):
    for pn in res[True][b][1]:
        a, c, d2 = res[True][b][1][pn], res[False][b][1][pn], res["double"][b][1][pn]
        sc = max(1.0, np.abs(c).max())
        print("batch %d %-48s scale %9.3g  fused-torch: max %9.3g frac>1e-4 %.4f relnorm %.2e | torch-torch (a second run): max %9.3g frac %.4f" % (
            b, pn, sc, np.abs(a - c).max(), (np.abs(a - c) > 1e-4 * sc).mean(), np.linalg.norm(a - c) / max(1.0, np.linalg.norm(c)),
            np.abs(d2 - c).max(), (np.abs(d2 - c) > 1e-4 * sc).mean()))
