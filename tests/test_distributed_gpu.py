"""Sharded (multi-rank) execution must equal single-rank execution bit for bit.
Two ranks share the one GPU of the test box (gloo rendezvous, exchange staged
through the host); on a real node the same code runs one rank per GPU over RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import inputs as I

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(rank, world, port, cfg, out):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), here, os.path.join(here, "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from helpers import build_tgn
    from zebra_amd.distributed import ShardedTGN, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        N, E, D, F, T, k, al, be, seed, bs = cfg[:10]
        strategy = cfg[10] if len(cfg) > 10 else "streaming"
        group = cfg[11] if len(cfg) > 11 else 0                # > 0: through the native pipeline, T-PPR launches over `group` batches
        native = cfg[12] if len(cfg) > 12 else None            # "msg" / "nomsg": the exchange INSIDE the native batch loop
                                                               # (TGN.enable_exchange, shared-memory transport: ranks share a GPU)
        src, dst, neg, ts, eidx = I.make_stream("bipartite" if strategy == "streaming" else "general", N, E, seed)
        w = I.model_weights(D, F, T, len(al), seed)
        _, efeat = I.random_tables(N, E + 1, D, F, seed)
        nf = None
        if strategy == "pruning":
            import types
            from zebra_amd.tppr import get_neighbor_finder
            nf = get_neighbor_finder(types.SimpleNamespace(sources=src, destinations=dst, edge_idxs=eidx, timestamps=ts))
        tgn = build_tgn(N, E + 1, D, F, T, k, al, be, w, efeat, strategy=strategy, nf=nf).eval()
        if world > 1 and strategy == "streaming":        # ranks share one GPU here: hub chains need a whole grid resident
            tgn.embedding_module.tppr_finder.set_device_share(world)
        runner = ShardedTGN(tgn, rank, world) if world > 1 else tgn
        dev = torch.device("cuda")
        t = [torch.from_numpy(x).to(dev) for x in (src, dst, neg, ts, eidx)]
        embs = []
        batches = [tuple(x[b * bs:min(E, (b + 1) * bs)] for x in t) for b in range((E + bs - 1) // bs)]   # (a ragged last batch where E % bs)
        if group and world > 1:
            tgn.enable_pipeline(tppr_cus=0, max_batch=bs, group=group)
        main = (getattr(tgn, "main_stream", None) if group and world > 1 else None) or torch.cuda.current_stream()
        if native and world > 1:
            # zt_pipeline_run shards every batch by (rank, world) and ends every step with pack -> all-gather -> scatter ->
            # projected-row refresh itself: no Python, no torch.distributed between two steps
            tgn.enable_exchange(rank, world, transport="shm", with_messages=(native == "msg"), shm_name="zt_test_%d" % port)
            H = D * (len(al) + 1)
            rows_max = max(shard_range(3 * len(c[0]), rank, world)[1] - shard_range(3 * len(c[0]), rank, world)[0] for c in batches)
            o = torch.zeros((len(batches), rows_max, H), dtype=torch.float32, device=dev)
            with torch.cuda.stream(main):
                tgn.run_device(tgn.prepare_run(batches), out=o)
            torch.cuda.synchronize()
            for b, c in enumerate(batches):
                lo, hi = shard_range(3 * len(c[0]), rank, world)
                embs.append(o[b, : hi - lo].cpu().numpy())
        with torch.cuda.stream(main):
            for b, cur in enumerate(batches):
                if native and world > 1:
                    break
                if group and world > 1:
                    embs.append(runner.step_device(*cur, ahead=batches[b + 1: b + 3 * group]).cpu().numpy())
                else:
                    embs.append(runner.step_device(*cur).cpu().numpy())
        torch.cuda.synchronize()
        if group and world > 1:
            tgn.enable_pipeline(False)
        state = {}
        if strategy == "streaming":
            tgn.embedding_module.tppr_finder.check_status()
            state = tgn.embedding_module.tppr_finder.export_state(0)
        m = tgn.memory
        out[rank] = dict(emb=embs, memory=m.memory.cpu().numpy(), last_update=m.last_update.cpu().numpy(),
                         messages=m.messages.cpu().numpy(), ts=m.timestamps.cpu().numpy(), flags=m.nodes.copy(),
                         state=state, shards=[shard_range(3 * len(c[0]), rank, world) for c in batches])
    finally:
        if world > 1:
            dist.destroy_process_group()


CFGS = {
    "streaming": (600, 2400, 100, 4, 100, 20, [0.1, 0.1], [0.5, 0.95], 301, 200),
    # config C4's shape (SuperUser: pruning T-PPR, k=40, F=1, width 10, depth 2), 4-GPU config in BASELINE.json
    "pruning_c4": (900, 3000, 100, 1, 100, 40, [0.1, 0.1], [0.5, 0.95], 302, 500, "pruning"),
    # the bench's multi-GPU path: ShardedTGN over the native pipeline, T-PPR launches over two batches
    "streaming_pipe": (600, 2400, 100, 4, 100, 20, [0.1, 0.1], [0.5, 0.95], 301, 200, "streaming", 2),
    "streaming_pipe1": (600, 2400, 100, 4, 100, 20, [0.1, 0.1], [0.5, 0.95], 301, 200, "streaming", 1),
    "pruning_pipe": (900, 3000, 100, 1, 100, 40, [0.1, 0.1], [0.5, 0.95], 302, 500, "pruning", 1),
    # 3 * bs not divisible by the world size and a shorter last batch: the rows a batch was queried ahead for are not the
    # rows its own step asks for (round-2 advisor), and the view ahead ends inside a group
    "pruning_pipe_uneven": (900, 2890, 100, 1, 100, 40, [0.1, 0.1], [0.5, 0.95], 303, 250, "pruning", 1),
    "streaming_pipe_uneven": (600, 2390, 100, 4, 100, 20, [0.1, 0.1], [0.5, 0.95], 304, 250, "streaming", 3),
    # the exchange inside the native batch loop (round 5): message rows in the payload, so that table can be compared too
    "streaming_native": (600, 2390, 100, 4, 100, 20, [0.1, 0.1], [0.5, 0.95], 305, 250, "streaming", 2, "msg"),
    "pruning_native": (900, 2890, 100, 1, 100, 40, [0.1, 0.1], [0.5, 0.95], 306, 250, "pruning", 1, "msg"),
    # ... and the payload of a real run: [id | memory row | last_update] only
    "streaming_native_nomsg": (600, 2400, 100, 172, 100, 20, [0.1, 0.1], [0.5, 0.95], 307, 200, "streaming", 2, "nomsg"),
}


@pytest.mark.parametrize("world,cfg_name", [(2, "streaming"), (3, "streaming"), (2, "pruning_c4"), (4, "pruning_c4"),
                                            (2, "streaming_pipe1"), (2, "streaming_pipe"), (3, "pruning_pipe"),
                                            (4, "pruning_pipe_uneven"), (4, "streaming_pipe_uneven"),
                                            (2, "streaming_native"), (3, "streaming_native"), (4, "pruning_native"),
                                            (2, "streaming_native_nomsg")])
def test_sharded_equals_single(world, cfg_name):
    cfg = CFGS[cfg_name]
    mgr = mp.get_context("spawn").Manager()      # never fork a process that holds GPU handles
    ref = mgr.dict()
    mp.spawn(_run, args=(1, _free_port(), cfg, ref), nprocs=1, join=True)
    out = mgr.dict()
    mp.spawn(_run, args=(world, _free_port(), cfg, out), nprocs=world, join=True)
    single = ref[0]
    for r in range(world):
        o = out[r]
        for b, e in enumerate(o["emb"]):
            lo, hi = o["shards"][b]
            assert np.array_equal(e, single["emb"][b][lo:hi]), "rank %d batch %d embeddings differ" % (r, b)
        # (the payload of a real run leaves the message rows out: an eval step never reads another rank's messages)
        tabs = ("memory", "last_update", "flags") if cfg_name.endswith("_nomsg") else ("memory", "last_update", "messages", "ts", "flags")
        for kk in tabs:
            assert np.array_equal(o[kk], single[kk]), "rank %d %s differs from the single-GPU run" % (r, kk)
        for kk in o["state"]:
            assert np.array_equal(o["state"][kk], single["state"][kk])


def test_pack_scatter_rows_match_torch_form():
    """zt_pack_rows / zt_scatter_rows (the GPU form of the exchange payload) against the torch form
    the CPU tests cover: same bytes out, same tables after the scatter."""
    from zebra_amd.distributed import pack_rows, unpack_rows, pack_rows_device, unpack_rows_device
    g = torch.Generator().manual_seed(5)
    N, D, msg, cap = 300, 100, 301, 64
    tabs = [torch.randn((N, D), generator=g), torch.randn(N, generator=g), torch.randn((N, msg), generator=g),
            torch.randn(N, generator=g)]
    ids = torch.randperm(N, generator=g)[:cap].to(torch.int32)
    for nv in (0, 1, 37, cap):
        n_valid = torch.tensor(nv, dtype=torch.int32)
        want = pack_rows(tabs, ids, n_valid, cap)
        dt = [t.cuda() for t in tabs]
        got = pack_rows_device(dt, ids.cuda(), n_valid.cuda(), cap)
        assert torch.equal(got.cpu().view(torch.int32), want.view(torch.int32))
        # scatter into fresh tables on both sides
        a = [torch.zeros_like(t) for t in tabs]
        b = [torch.zeros_like(t).cuda() for t in tabs]
        recv = torch.cat([want, want.flip(0)])          # every row twice, like two ranks sending
        unpack_rows(a, recv)
        unpack_rows_device(b, recv.cuda().contiguous())
        torch.cuda.synchronize()
        for x, y in zip(a, b):
            assert torch.equal(x, y.cpu())


def _train_run(rank, world, port, out):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), here, os.path.join(here, "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from helpers import build_tgn
    from zebra_amd.distributed import ShardedTGN
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        N, E, D, F, T, k, al, be, seed, bs = 300, 640, 100, 4, 100, 20, [0.1, 0.1], [0.5, 0.95], 303, 160
        src, dst, neg, ts, eidx = I.make_stream("bipartite", N, E, seed)
        w = I.model_weights(D, F, T, len(al), seed)
        _, efeat = I.random_tables(N, E + 1, D, F, seed)
        tgn = build_tgn(N, E + 1, D, F, T, k, al, be, w, efeat)
        if world > 1:
            tgn.embedding_module.tppr_finder.set_device_share(world)
        tgn.train(True)
        crit = torch.nn.BCELoss()
        runner = ShardedTGN(tgn, rank, world)
        res = []
        for b in range(E // bs):
            s, e = b * bs, (b + 1) * bs
            tgn.zero_grad()
            if world > 1:
                loss = runner.train_step(src[s:e], dst[s:e], neg[s:e], ts[s:e], eidx[s:e], 10, crit)
                lt = loss.clone().cpu()
                dist.all_reduce(lt)                                   # shares of the loss add up
                loss = float(lt)
            else:
                pos, negp = tgn.compute_edge_probabilities(src[s:e], dst[s:e], neg[s:e], ts[s:e], eidx[s:e], 10, True)
                l = crit(pos.squeeze(), torch.ones(bs, device="cuda")) + crit(negp.squeeze(), torch.zeros(bs, device="cuda"))
                l.backward()
                loss = float(l)
            res.append((loss, {pn: p.grad.detach().cpu().numpy().copy() for pn, p in tgn.named_parameters() if p.grad is not None}))
        out[rank] = dict(steps=res, memory=tgn.memory.memory.cpu().numpy())
    finally:
        if world > 1:
            dist.destroy_process_group()


def test_data_parallel_training_step_equals_single():
    """two ranks, each embedding and scoring half of the batch's edges, gradients summed with
    the bucketed all-reduce (gloo here, RCCL on a node): loss and all parameter gradients of four dependent
    training steps equal the single-process step; the replicated state stays identical."""
    mgr = mp.get_context("spawn").Manager()      # never fork a process that holds GPU handles
    ref, out = mgr.dict(), mgr.dict()
    mp.spawn(_train_run, args=(1, _free_port(), ref), nprocs=1, join=True)
    mp.spawn(_train_run, args=(2, _free_port(), out), nprocs=2, join=True)
    for r in range(2):
        assert np.array_equal(out[r]["memory"], ref[0]["memory"])
        for b, ((la, ga), (lb, gb)) in enumerate(zip(out[r]["steps"], ref[0]["steps"])):
            assert abs(la - lb) <= 1e-5, "loss of step %d" % b
            assert set(gb) <= set(ga)
            for pn in ga:
                if pn not in gb:                      # never touched by the loss: the all-reduce filled in zeros
                    assert not ga[pn].any(), pn
                    continue
                assert np.linalg.norm(ga[pn] - gb[pn]) <= 2e-3 * max(1e-3, np.linalg.norm(gb[pn])), "%s step %d" % (pn, b)


def _nccl_run(rank, port, out):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), here, os.path.join(here, "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from helpers import build_tgn
    from zebra_amd.distributed import ShardedTGN
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        N, E, D, F, T, k, al, be, seed, bs = 600, 2400, 100, 4, 100, 20, [0.1, 0.1], [0.5, 0.95], 301, 200
        src, dst, neg, ts, eidx = I.make_stream("bipartite", N, E, seed)
        w = I.model_weights(D, F, T, 2, seed)
        _, efeat = I.random_tables(N, E + 1, D, F, seed)
        dev = torch.device("cuda")
        t = [torch.from_numpy(x).to(dev) for x in (src, dst, neg, ts, eidx)]
        batches = [tuple(x[b * bs:(b + 1) * bs] for x in t) for b in range(E // bs)]
        res = {}
        for mode in ("plain", "sharded", "native"):
            tgn = build_tgn(N, E + 1, D, F, T, k, al, be, w, efeat).eval()
            embs = []
            if mode == "plain":
                for cur in batches:
                    embs.append(tgn.step_device(*cur).cpu().numpy())
            elif mode == "native":
                # the exchange inside the library's batch loop: an RCCL communicator made by the library itself (world 1 is
                # what one GPU allows), ncclAllGather on the CU-masked main stream
                tgn.enable_pipeline(tppr_cus=64, max_batch=bs, group=2)
                tgn.enable_exchange(0, 1, transport="rccl")
                o = torch.zeros((len(batches), 3 * bs, D * 3), dtype=torch.float32, device=dev)
                with torch.cuda.stream(tgn.main_stream):
                    tgn.run_device(tgn.prepare_run(batches), out=o)
                torch.cuda.synchronize()
                embs = [o[b].cpu().numpy() for b in range(len(batches))]
                tgn.enable_pipeline(False)
            else:
                tgn.enable_pipeline(tppr_cus=64, max_batch=bs, group=2)
                runner = ShardedTGN(tgn, 0, 1)
                with torch.cuda.stream(tgn.main_stream):
                    for b, cur in enumerate(batches):
                        embs.append(runner.step_device(*cur, ahead=batches[b + 1:b + 6]).cpu().numpy())
                torch.cuda.synchronize()
                tgn.enable_pipeline(False)
            res[mode] = (embs, tgn.memory.memory.cpu().numpy())
        out[0] = res
    finally:
        dist.destroy_process_group()


def test_rccl_exchange_on_cu_masked_streams():
    """What one GPU can show of the multi-GPU bench path WITH RCCL: a process group of one rank over the "nccl"
    backend, ShardedTGN over the native pipeline on CU-masked streams (the all-gather of touched rows runs on
    torch.distributed's own stream, ordered against the masked main stream by events).  Equals the plain run."""
    mgr = mp.get_context("spawn").Manager()
    out = mgr.dict()
    mp.spawn(_nccl_run, args=(_free_port(), out), nprocs=1, join=True)
    a, b, c = out[0]["plain"], out[0]["sharded"], out[0]["native"]
    assert all(np.array_equal(x, y) for x, y in zip(a[0], b[0]))
    assert np.array_equal(a[1], b[1])
    assert all(np.array_equal(x, y) for x, y in zip(a[0], c[0]))
    assert np.array_equal(a[1], c[1])


@pytest.mark.parametrize("workload", ["c2", "c5"])
def test_bench_gpus_2_rehearsal(workload):
    """`python bench.py --gpus 2` end to end on ONE GPU (ZT_BENCH_REHEARSAL=1: both ranks on cuda:0; every rank's T-PPR
    launches take half the stream's CUs and run without hub chains -- zt_tppr_set_device_share --; the row exchange inside
    the native step loop goes through shared memory, RCCL refusing two ranks on one device): the parent starts the two
    ranks, the sharded native pipeline runs, rank 0's line says n_gpus = 2.  C2 with a short prefill; C5 as the default
    command runs it -- CU masks on, the full 10 % prefill (2 441 batches), 10 000 001 nodes: the layout whose round-4
    rehearsal ended in a dependency-wait time-out.  (What the driver's scaling run does with RCCL on
    separate GPUs.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["ZT_BENCH_REHEARSAL"] = "1"
    extra = ["--prefill-steps", "20"] if workload == "c2" else ["--legs", "none", "--no-score"]
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workload", workload, "--steps", "6",
                        "--warmup", "2", "--cpu-edges", "0"] + extra, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 6 and out["value"] > 0
    assert out["config"]["step_loop"].startswith("native")
