"""world_size-2 gloo tests of the multi-GPU exchange logic (zebra_amd.distributed)
on CPU tensors: sharding arithmetic and the fixed-size all-gather of touched
rows.  The GPU kernels are not involved (no GPU in the build container)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from zebra_amd.distributed import exchange_touched_rows, pack_rows, shard_capacity, shard_range, unpack_rows


def test_shard_ranges_partition_everything():
    for n in (0, 1, 7, 600, 8192, 12288):
        for world in (1, 2, 3, 4, 8):
            cover = []
            for r in range(world):
                lo, hi = shard_range(n, r, world)
                assert 0 <= lo <= hi <= n
                cover += list(range(lo, hi))
            assert cover == list(range(n))
            assert shard_capacity(n, world) >= (n + world - 1) // world - 0


def test_pack_unpack_roundtrip_single_process():
    N, D, W = 50, 6, 9
    rng = np.random.RandomState(0)
    mem = torch.from_numpy(rng.standard_normal((N, D)).astype(np.float32))
    lu = torch.from_numpy(rng.standard_normal(N).astype(np.float32))
    msg = torch.from_numpy(rng.standard_normal((N, W)).astype(np.float32))
    ids = torch.tensor([7, 3, 41, 0, 0, 0], dtype=torch.int32)
    buf = pack_rows([mem, lu, msg], ids, torch.tensor([3], dtype=torch.int32), cap=6)
    assert buf.shape == (6, 1 + D + 1 + W)
    got = buf[:, 0].contiguous().view(torch.int32).tolist()
    assert got == [7, 3, 41, -1, -1, -1]
    mem2, lu2, msg2 = torch.zeros_like(mem), torch.zeros_like(lu), torch.zeros_like(msg)
    assert unpack_rows([mem2, lu2, msg2], buf) == 3
    for v in (7, 3, 41):
        assert torch.equal(mem2[v], mem[v]) and lu2[v] == lu[v] and torch.equal(msg2[v], msg[v])
    assert mem2[0].abs().sum() == 0      # padding entries were not written


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out, shape=(200, 8, 11, 32)):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        N, D, W, B = shape
        rng = np.random.RandomState(5)                       # same on every rank: replicated tables
        mem = torch.from_numpy(rng.standard_normal((N, D)).astype(np.float32))
        lu = torch.zeros(N)
        msg = torch.zeros((N, W))
        # "batch": 2B endpoint positions; winner of a node = its last position
        ends = torch.from_numpy(rng.randint(1, N, 2 * B).astype(np.int64))
        last = {int(v): p for p, v in enumerate(ends.tolist())}
        lo, hi = shard_range(2 * B, rank, world)
        mine = sorted(v for v, p in last.items() if lo <= p < hi)
        cap = shard_capacity(2 * B, world)
        assert len(mine) <= cap
        # local "update" of my winners (a deterministic function of the id)
        for v in mine:
            mem[v] = torch.full((D,), float(v))
            lu[v] = v + 0.5
            msg[v] = torch.arange(W, dtype=torch.float32) + v
        ids = torch.zeros(cap, dtype=torch.int32)
        ids[: len(mine)] = torch.tensor(mine, dtype=torch.int32)
        n = exchange_touched_rows([mem, lu, msg], ids, torch.tensor([len(mine)], dtype=torch.int32), cap)
        assert n == len(last)
        # every replica must now hold every winner's row
        vs = torch.tensor(sorted(last), dtype=torch.int64)
        assert torch.equal(mem[vs], vs.to(torch.float32).reshape(-1, 1).expand(-1, D))
        assert torch.equal(lu[vs], vs.to(torch.float32) + 0.5)
        assert torch.equal(msg[vs], torch.arange(W, dtype=torch.float32).reshape(1, -1) + vs.to(torch.float32).reshape(-1, 1))
        # replicas are identical
        digest = torch.cat([mem.reshape(-1), lu, msg.reshape(-1)])
        gathered = [torch.empty_like(digest) for _ in range(world)]
        dist.all_gather(gathered, digest)
        assert all(torch.equal(g, gathered[0]) for g in gathered)
        out[rank] = 1
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,shape", [(2, (200, 8, 11, 32)), (3, (200, 8, 11, 32)),
                                         (8, (20000, 100, 301, 4096)),      # C5's step: 8 192 positions, rows of 100 + 1 + 301 floats
                                         (8, (20000, 100, 301, 4059))])     # ... and a ragged last batch (positions not divisible by 8)
def test_exchange_touched_rows_gloo(world, shape):
    """The touched-row exchange of a sharded step over gloo: world 2 and 3 on a toy shape, world 8 at
    C5's sizes -- every rank owns the winners at its slice of the 2B positions, one fixed-size all-gather, every replica
    ends up with every winner's rows and all replicas are identical."""
    mgr = mp.get_context("spawn").Manager()      # never fork a process that holds GPU handles
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out, shape), nprocs=world, join=True)
    assert sorted(out.keys()) == list(range(world))


def _grads_worker(rank, world, port, out):
    import torch.distributed as dist
    from zebra_amd.distributed import allreduce_gradients
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        model = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.ReLU(), torch.nn.Linear(5, 1))
        frozen = torch.nn.Parameter(torch.ones(3), requires_grad=False)
        x = torch.arange(28, dtype=torch.float32).reshape(4, 7) / 10 + rank
        model(x).sum().backward()
        if rank == 1:
            model[2].bias.grad = None                        # a parameter unused on one rank still takes part
        local = [None if p.grad is None else p.grad.clone() for p in model.parameters()]
        gathered = [None] * world
        dist.all_gather_object(gathered, [None if g is None else g.numpy() for g in local])
        n = allreduce_gradients(list(model.parameters()) + [frozen], bucket_bytes=64)     # several small buckets
        assert n >= 2
        for q, p in enumerate(model.parameters()):
            want = sum(torch.from_numpy(g[q]) for g in gathered if g[q] is not None)
            assert torch.allclose(p.grad, want, atol=1e-6), q
        out[rank] = 1
    finally:
        dist.destroy_process_group()


def test_allreduce_gradients_gloo():
    """The bucketed gradient all-reduce of data-parallel training (on GPUs: RCCL) with world_size 2 on gloo."""
    import torch.multiprocessing as mp
    mgr = mp.get_context("spawn").Manager()      # never fork a process that holds GPU handles
    out = mgr.dict()
    mp.spawn(_grads_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert len(out) == 2


def test_bench_gpus_flag_starts_that_many_ranks():
    """`python bench.py --gpus 2` (no launcher around it, the form the driver uses for N = 1) must start two
    ranks itself -- the parent as a plain child-process launcher that never touches the GPU -- and rank 0's
    line must say n_gpus = 2 (round-2 review: --gpus was parsed and never read).  --dry-run: rendezvous
    over gloo only, so this runs without a GPU."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["launched_by"] == "bench.py"
    # a launcher that started a different number of ranks than --gpus says is an error, not a silent 1-GPU run
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"], env=env2,
                        capture_output=True, text=True, timeout=300)
    assert r2.returncode != 0
