# where does the sharded + pipelined run diverge from the sharded run without the pipeline?  (2 ranks on one GPU, gloo)
import os, sys, socket
import numpy as np, torch, torch.distributed as dist, torch.multiprocessing as mp
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/tests/golden')
import inputs as I

def run(rank, world, port, pipe, out):
    from helpers import build_tgn
    from zebra_amd.distributed import ShardedTGN
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ZT_STREAM_CHAINS="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    N, E, D, F, T, k, al, be, seed, bs = 600, 1200, 100, 4, 100, 20, [0.1, 0.1], [0.5, 0.95], 301, 200
    src, dst, neg, ts, eidx = I.make_stream("bipartite", N, E, seed)
    w = I.model_weights(D, F, T, 2, seed)
    _, efeat = I.random_tables(N, E + 1, D, F, seed)
    tgn = build_tgn(N, E + 1, D, F, T, k, al, be, w, efeat).eval()
    runner = ShardedTGN(tgn, rank, world)
    dev = torch.device("cuda")
    t = [torch.from_numpy(x).to(dev) for x in (src, dst, neg, ts, eidx)]
    batches = [tuple(x[b * bs:(b + 1) * bs] for x in t) for b in range(E // bs)]
    if pipe:
        tgn.enable_pipeline(tppr_cus=0, max_batch=bs, group=1)
    main = (getattr(tgn, "main_stream", None) if pipe else None) or torch.cuda.current_stream()
    res = []
    with torch.cuda.stream(main):
        for b, cur in enumerate(batches):
            emb = runner.step_device(*cur, ahead=batches[b + 1:b + 3] if pipe == 2 else None)
            torch.cuda.synchronize()
            res.append((emb.cpu().numpy(), tgn.memory.memory.cpu().numpy().copy(), tgn.memory.last_update.cpu().numpy().copy()))
    out[rank] = res
    if pipe:
        tgn.enable_pipeline(False)
    dist.destroy_process_group()

def port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p

if __name__ == "__main__":
    mgr = mp.get_context("spawn").Manager()
    outs = {}
    for pipe in (0, 1, 2):
        o = mgr.dict()
        mp.spawn(run, args=(2, port(), pipe, o), nprocs=2, join=True)
        outs[pipe] = {r: o[r] for r in range(2)}
    for pipe in (1, 2):
        for b in range(len(outs[0][0])):
            for r in range(2):
                e0, m0, l0 = outs[0][r][b]; e1, m1, l1 = outs[pipe][r][b]
                dm = np.where(np.abs(m0 - m1).max(axis=1) > 0)[0]
                print("pipe=%d batch %d rank %d: emb equal %s, memory rows differing %d %s, last_update differing %d" % (
                    pipe, b, r, np.array_equal(e0, e1), len(dm), dm[:8], int((l0 != l1).sum())))
