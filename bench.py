#!/usr/bin/env python3
"""bench.py -- temporal edges/sec embedded by the T-PPR + top-k aggregate +
memory-update path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c5|c2|c3|c4]

One "step" = one pass of the hot path over one batch of the synthetic stream
(eval-mode protocol of reference model/tgn_model.py:124-174): streaming (or
pruning) T-PPR update + row emission, gather/TimeEncode/transform/weighted
sum for the 3B rows x n_tppr models, last-message store and GRU memory update.
All inputs are resident in HBM when the timed region starts.

N > 1 runs one rank per GPU over RCCL.  Either a launcher starts the ranks
(`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`:
RANK / LOCAL_RANK / WORLD_SIZE in the environment), or `python bench.py --gpus N`
starts them itself: the parent -- BEFORE it has touched the GPU -- runs that
launcher as a child process, lets rank 0's JSON line through and exits with
the children's status.  T-PPR state and memory are replicated, every rank
applies the whole batch's T-PPR update (bit-identical replicas, no
communication), the 3B embedding rows and the touched-endpoint memory updates
are sharded, and the touched memory rows are exchanged with one all-gather per
batch.  Total work per step is fixed ("strong" scaling).
`--dry-run` stops after the rendezvous (no GPU needed: the CPU-side test of the
launch path).

Rank 0 prints ONE JSON line (contract in the task statement) with the
`roofline` of the dominant kernel and the `cpu_baseline` (oracle timed on the
host cores, N = 1 only).
"""
import argparse
import contextlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

_TORCH_DEFAULT_THREADS = None
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3   # dense f32 MFMA


def algorithmic_bytes(k, F, M, D=100, T=100, strategy="streaming", width=10, depth=2):
    """logical bytes per edge, no cache credit."""
    msg = 2 * D + F + T
    if strategy == "streaming":
        p1 = 5 * (24 * k + 8) + 48 * k                # per model: 3 rows read, 2 written, 3 output rows
    else:                                             # pruning: CSR tails of width + width^2 + ... entries x 16 B
        states = sum(width ** d for d in range(1, depth + 1))
        p1 = 3 * states * 16 + 48 * k                 # per model (3 query rows per edge)
    p2 = 3 * k * (4 * D + 4 * F) + 3 * k * 16         # per model
    p2_once = 3 * 4 * D + 3 * 4 * D * (M + 1)
    p3 = 2 * (4 * msg + 4 * D + 4 * D + 4) + 2 * (4 * msg + 4)
    return dict(p1=p1, p2=p2, p2_once=p2_once, p3=p3, total=M * (p1 + p2) + p2_once + p3)


def algorithmic_flops(k, F, M, D=100, T=100):
    """FLOPs per edge as the reference computes them (fc2 per neighbour), scorer included."""
    msg = 2 * D + F + T
    H = D * (M + 1)
    p2 = 3 * k * (2 * (D + F + T) * D + 2 * D * D)    # per model (reference formulation)
    src = 3 * 4 * D * D
    gru = 2 * 2 * 3 * D * (msg + D)
    scorer = 2 * (2 * (2 * H) * H + 2 * H)
    return dict(p2=p2, src=src, gru=gru, scorer=scorer, total=M * p2 + src + gru + scorer)


def executed_flops(k, F, M, D=100, T=100, projected=True):
    """FLOPs the HIP kernels execute per edge: fc2 is hoisted behind the k-reduction (aggregate.hip), so
    k_fc1_agg runs fc1 only and k_embed_out runs fc2 once per row and model plus the source transform.
    With the projected memory table (W_m memory[v] kept per node) k_fc1_agg contracts over the F + T
    edge-feature / time columns only, and k_project_rows redoes W_m memory[v] for the <= 2 rows per edge the
    GRU rewrites."""
    kc = (F + T) if projected else (D + F + T)
    return dict(fc1_agg=M * 3 * k * 2 * kc * D, embed_out=M * 3 * 2 * D * D + 3 * 4 * D * D,
                project_rows=M * 2 * 2 * D * D if projected else 0)


def make_stream(wl, n_edges, seed=2020, perm_seed=None):
    from zebra_amd import synth
    src, dst, ts, eidx = synth.power_law_stream(wl["n_nodes"], n_edges, bipartite=wl["bipartite"], seed=seed,
                                                perm_seed=perm_seed)
    neg = synth.negatives(dst, n_edges, seed=seed + 1)
    return src, dst, neg, ts, eidx


def build_model(wl, device, n_edge_rows):
    import types

    import torch
    from zebra_amd.tgn import TGN
    from zebra_amd.tppr import get_neighbor_finder
    torch.manual_seed(0)
    args = types.SimpleNamespace(alpha_list=list(wl["alpha"]), beta_list=list(wl["beta"]), topk=wl["k"],
                                 tppr_strategy=wl["strategy"], n_degree=wl.get("width", 10),
                                 n_layer=wl.get("depth", 2), n_nodes=wl["n_nodes"] + 1, n_edges=n_edge_rows)
    F = wl["F"]
    if F == 1:
        efeat = torch.zeros((n_edge_rows, 1), dtype=torch.float32, device=device)
    else:
        g = torch.Generator(device="cpu").manual_seed(1)
        efeat = torch.randn((n_edge_rows, F), generator=g, dtype=torch.float32)
        efeat[0] = 0
        efeat = efeat.to(device)
    tgn = TGN(neighbor_finder=None, node_features=None, edge_features=efeat, device=device, n_layers=args.n_layer,
              use_memory=True, embedding_module_type="diffusion", message_function="identity",
              memory_updater_type="gru", n_neighbors=args.n_degree, args=args).to(device).eval()
    return tgn


def snapshot_state(tgn, wl, touched):
    """Host copy of the warm state (T-PPR rows, memory tables) of the nodes the stream has touched so far:
    what the CPU baseline starts from, so that both legs see the same warm-up."""
    import torch
    ids = np.ascontiguousarray(touched, np.int64)
    snap = {"ids": ids}
    if wl["strategy"] == "streaming":
        f = tgn.embedding_module.tppr_finder
        snap["tppr"] = [f.export_rows(m, ids) for m in range(f.n_tppr)]
    ids_d = torch.from_numpy(ids).to(tgn.device)
    m = tgn.memory
    for name in ("memory", "last_update", "messages", "timestamps"):
        snap[name] = getattr(m, name).index_select(0, ids_d).cpu().numpy()
    snap["flags"] = m.flags.index_select(0, ids_d).cpu().numpy()
    return snap


def model_weights(tgn):
    em, g = tgn.embedding_module, tgn.memory_updater.memory_updater
    c = lambda t: t.detach().cpu().numpy().copy()
    w = dict(fc1_w=c(em.fc1.weight), fc1_b=c(em.fc1.bias), fc2_w=c(em.fc2.weight), fc2_b=c(em.fc2.bias),
             fc1s_w=c(em.fc1_source.weight), fc1s_b=c(em.fc1_source.bias), fc2s_w=c(em.fc2_source.weight),
             fc2s_b=c(em.fc2_source.bias), w_ih=c(g.weight_ih), w_hh=c(g.weight_hh), b_ih=c(g.bias_ih),
             b_hh=c(g.bias_hh), aff1_w=c(tgn.affinity_score.fc1.weight), aff1_b=c(tgn.affinity_score.fc1.bias),
             aff2_w=c(tgn.affinity_score.fc2.weight), aff2_b=c(tgn.affinity_score.fc2.bias))
    return w, c(tgn.time_encoder.w.weight).ravel()


def cpu_baseline(wl, snap, weights, time_w, batches, csr_arrays, n_threads, gpu=None):
    """Oracle ("port") on the host cores: the eval-mode protocol on the first timed batches of the same
    stream, started from the GPU run's warm state.  SURVEY.md 8(d): P1 = the C restatement, single-threaded
    like the reference's Numba loop, P2 / P3 = the build's torch-CPU module (oracle/torch_cpu.py: the torch
    ops the reference runs, torch.get_num_threads() threads) -- that is `value`, first half of the sample.
    Second half, reported beside it: P1 with one thread per T-PPR model and P2 / P3 by the C port on
    n_threads OpenMP threads.
    ``gpu``: what the GPU run kept of the SAME batches (``emb(b)`` -> its [3B, H] embeddings of sample batch b as a numpy
    array; ``after`` = ids + T-PPR rows + memory tables exported right behind the timed region, present when the sample
    ends where the timed region ends): the oracle's outputs are compared with them batch by batch instead of being
    thrown away -> ``parity_in_run`` (the oracle is the checker here, after the timed regions)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle
    import torch
    global _TORCH_DEFAULT_THREADS
    if _TORCH_DEFAULT_THREADS is None:
        _TORCH_DEFAULT_THREADS = torch.get_num_threads()
    default_threads = _TORCH_DEFAULT_THREADS
    torch.set_num_threads(n_threads)
    bs, k, M, F = wl["bs"], wl["k"], len(wl["alpha"]), wl["F"]
    D = T = 100
    N = wl["n_nodes"] + 1
    efeat = np.zeros((wl["n_edges"] + 1, 1), np.float32) if F == 1 else csr_arrays["efeat"]
    finder = None
    if wl["strategy"] == "pruning":
        finder = pyoracle.CsrOracle(csr_arrays["src"], csr_arrays["dst"], csr_arrays["eidx"], csr_arrays["ts"], N)
    p = pyoracle.ProtocolOracle(N, D, F, T, k, wl["alpha"], wl["beta"], weights, efeat, time_w, wl["strategy"], finder,
                                wl.get("width", 10), wl.get("depth", 2), n_threads=n_threads)
    ids = snap["ids"]
    if wl["strategy"] == "streaming":
        for m in range(M):
            p.tppr.import_rows(m, ids, snap["tppr"][m])
    p.mem.memory[ids] = snap["memory"]
    p.mem.last_update[ids] = snap["last_update"]
    p.mem.messages[ids] = snap["messages"]
    p.mem.timestamps[ids] = snap["timestamps"]
    p.mem.flags[ids] = snap["flags"]
    p.test_mode = True                              # the GPU leg flushed pending messages before its warm-up
    # three parts of the sample: (0) torch-CPU P2/P3 on n_threads threads = `value`; (1) the C port on n_threads OpenMP
    # threads, P1 with one thread per model; (2) torch-CPU P2/P3 on torch's DEFAULT thread count for this host (what
    # `import torch` picks: the host's cores) -- reported beside `value`, since 16 threads are not what the host can do
    nb = len(batches)
    cut1, cut2 = max(1, (2 * nb) // 5), max(2, (4 * nb) // 5)
    t_p1 = [0.0, 0.0, 0.0]
    t_all = [0.0, 0.0, 0.0]
    n_e = [0, 0, 0]
    single = p.tppr.streaming_topk if p.tppr is not None else None
    par = dict(batches=0, timed_batches=0, max_abs_embedding_diff=0.0, tolerance=1e-4)
    for b, (src, dst, neg, ts, eidx) in enumerate(batches):
        part = 0 if b < cut1 else (1 if b < cut2 else 2)
        p.p23 = "c" if part == 1 else "torch"
        if part == 2 and b == cut2:
            torch.set_num_threads(default_threads)
            p._t23 = None                            # (the torch-CPU module is rebuilt for the new thread count)
            p.n_threads = default_threads
        if p.tppr is not None:
            p.tppr.streaming_topk = p.tppr.streaming_topk_threads if part == 1 else single
        t0 = time.perf_counter()
        nodes = np.concatenate([src, dst, neg]).astype(np.int32)
        on = p.topk(nodes, ts, eidx)
        t1 = time.perf_counter()
        emb_ref, _ = p.batch(src, dst, neg, ts, eidx, False, topk_out=on)
        t2 = time.perf_counter()
        t_p1[part] += t1 - t0
        t_all[part] += t2 - t0
        n_e[part] += len(src)
        if gpu is not None:
            got = gpu["emb"](b)
            if got is not None:
                d_b = float(np.abs(got - np.asarray(emb_ref)).max()) if got.shape == np.asarray(emb_ref).shape else float("inf")
                par["max_abs_embedding_diff"] = max(par["max_abs_embedding_diff"], d_b)
                par["batches"] += 1
                if b >= gpu["warmup"]:
                    par["timed_batches"] += 1
            del got
    if p.tppr is not None:
        p.tppr.streaming_topk = single
    out = dict(value=n_e[0] / t_all[0], unit="edges/s", cores=n_threads, kind="port",
               host_cpus=os.cpu_count(), torch_num_threads=n_threads,
               value_is="P1: C port on 1 thread; P2/P3: torch-CPU ops (oracle/torch_cpu.py)",
               p1_edges_per_s_1thread=n_e[0] / max(t_p1[0], 1e-9),
               p23_edges_per_s_torch_cpu=n_e[0] / max(t_all[0] - t_p1[0], 1e-9),
               sample="%d batches (%d edges) of the same stream starting from the GPU run's state after the prefill "
                      "(where its warm-up steps start); value = first %d batches: T-PPR loop (C port) on 1 thread as "
                      "the reference's Numba loop, aggregation / messages / GRU as torch-CPU ops on %d threads; the next %d: "
                      "T-PPR with one thread per model, aggregation / GRU by the C port on %d OpenMP threads; the last %d: as "
                      "the first with torch's default thread count (%d)"
                      % (nb, sum(n_e), cut1, n_threads, cut2 - cut1, n_threads, nb - cut2, default_threads))
    if n_e[1]:
        out["value_c_port"] = n_e[1] / t_all[1]
        out["c_port_threads"] = n_threads
        out["p1_threads"] = M if wl["strategy"] == "streaming" else 1
        out["p1_edges_per_s_threads"] = n_e[1] / max(t_p1[1], 1e-9)
        out["p23_edges_per_s_c_port"] = n_e[1] / max(t_all[1] - t_p1[1], 1e-9)
    if n_e[2]:
        out["value_default_threads"] = n_e[2] / t_all[2]
        out["default_threads"] = default_threads
    torch.set_num_threads(default_threads)
    if gpu is not None and par["batches"]:
        after = gpu.get("after")
        par["tppr_rows_bit_exact"] = None
        if after is not None:
            ids_a = after["ids"]
            if p.tppr is not None:
                same, n_rows = True, 0
                for m in range(M):
                    ref_rows = p.tppr.export_rows(m, ids_a)
                    for kk in ref_rows:
                        same = same and bool(np.array_equal(ref_rows[kk], after["tppr"][m][kk]))
                    n_rows += len(ids_a)
                par["tppr_rows_bit_exact"] = same
                par["tppr_rows_compared"] = n_rows
            par["max_abs_memory_diff"] = float(np.abs(p.mem.memory[ids_a] - after["memory"]).max()) if len(ids_a) else 0.0
            par["last_update_equal"] = bool(np.array_equal(p.mem.last_update[ids_a], after["last_update"]))
        par["ok"] = bool(par["max_abs_embedding_diff"] <= par["tolerance"] and par["tppr_rows_bit_exact"] is not False
                         and par.get("max_abs_memory_diff", 0.0) <= par["tolerance"] and par.get("last_update_equal", True))
        par["note"] = ("oracle (CPU restatement, pinned to the reference's fixtures) on the run's own batches from the run's own "
                       "warm state: %d warm-up + %d timed batches, embeddings of every batch against the ones the GPU wrote in "
                       "the warm-up / timed region%s" % (par["batches"] - par["timed_batches"], par["timed_batches"],
                       "; T-PPR rows (len, norm, keys, time stamps, float64 weights: np.array_equal) and memory rows of every "
                       "node those batches touched, exported right behind the timed region" if after is not None else
                       "; state not compared (the sample ends before the timed region does)"))
        out["parity_in_run"] = par
    return out


def launch_ranks(n, argv):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks as a CHILD process (the
    torch.distributed launcher) before this process has made any GPU call, let their output through
    (rank 0 prints the JSON line) and exit with their status.  Never exec: a process that has initialised
    the GPU must not be replaced, and this one stays clear of the GPU altogether."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["ZT_BENCH_LAUNCHED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    sys.stderr.write("[bench] --gpus %d: starting %d ranks: %s\n" % (n, n, " ".join(cmd)))
    return subprocess.run(cmd, env=env).returncode


CRIT_SECTION_NS = 970.0    # the hub chain's critical section: 2280 core clocks (median, tools/crit_profile.py)


def run_workload(a, name, steps, warmup, world, rank, device, headline, cpu_edges, steady=0):
    """One workload: build the model, prefill, warm up, time `steps` steps, read the per-kernel events, run the CPU leg
    (world == 1).  Returns the dict of the JSON line (headline) or of a `workloads` entry."""
    import ctypes as C

    import torch
    import torch.distributed as dist
    from zebra_amd import _capi, synth

    wl = dict(synth.WORKLOADS[name])
    bs, k, M, F = wl["bs"], wl["k"], len(wl["alpha"]), wl["F"]
    # the first 10 % of the stream is the untimed warm-up, so that T-PPR rows are full
    prefill = a.prefill_steps if (a.prefill_steps >= 0 and headline) else (wl["n_edges"] // 10) // bs
    # two timed regions: the metric's step (embeddings + T-PPR and memory updates: SURVEY.md 8d) and, behind it, the same
    # step with compute_edge_probabilities' scorer at its tail (model/tgn_model.py:185-188; 5 warm-up steps in between)
    scored = world == 1 and not a.no_score and not (a.no_pipeline and headline) and not a.exchange_world1
    # a third region (headline only, `steady` > 0): the metric's step again over `steady` batches with the launch groups such
    # a region gets -- the figure DESIGN.md quotes as "200 steps", where the region's first and last batch no longer count
    steady = steady if (steady > steps and not (a.no_pipeline and headline) and not a.python_loop) else 0
    n_steps_total = prefill + warmup + steps + ((5 + steps) if scored else 0) + ((10 + steady) if steady else 0)
    n_edges = n_steps_total * bs
    if n_edges > wl["n_edges"]:
        raise SystemExit("stream of %d edges is shorter than prefill+warmup+steps" % wl["n_edges"])

    src, dst, neg, ts, eidx = make_stream(wl, n_edges, perm_seed=None if a.perm_seed < 0 else a.perm_seed)
    n_edge_rows = (wl["n_edges"] if F == 1 else n_edges) + 1      # F=1: the full |E|+1 zero table is cheap
    tgn = build_model(wl, device, n_edge_rows)
    if world > 1 and os.environ.get("ZT_BENCH_REHEARSAL") == "1" and wl["strategy"] == "streaming":
        tgn.embedding_module.tppr_finder.set_device_share(world)
    if wl["strategy"] == "pruning":
        import types
        from zebra_amd.tppr import get_neighbor_finder
        tgn.set_neighbor_finder(get_neighbor_finder(types.SimpleNamespace(sources=src, destinations=dst,
                                                                          edge_idxs=eidx, timestamps=ts)))
    src_d = torch.from_numpy(src).to(device)
    dst_d = torch.from_numpy(dst).to(device)
    neg_d = torch.from_numpy(neg).to(device)
    ts_d = torch.from_numpy(ts).to(device)
    eidx_d = torch.from_numpy(eidx).to(device)

    if world > 1:
        from zebra_amd.distributed import ShardedTGN
        runner = ShardedTGN(tgn, rank, world)
        step = runner.step_device
    else:
        step = tgn.step_device
    rehearsal = os.environ.get("ZT_BENCH_REHEARSAL") == "1"

    # the launch configuration: ONE function, shared with tests/test_configs_gpu.py (which runs every BASELINE config at
    # its real shape against the oracle with exactly these settings)
    _capi.set_kernel_choice(_capi.CHOICE_GROUP_RELEASE, _capi.RELEASE_LAUNCH_FULL if a.release_by_launch_full else
                            (_capi.RELEASE_LAUNCH if a.release_by_launch else 0))
    for kv in a.choice:
        which, value = kv.split("=")
        _capi.set_kernel_choice(int(which), int(value))
    tppr_cus, group = synth.pipeline_settings(wl, steps, a.tppr_cus if headline else -1, a.group if headline else -1)
    no_pipeline = a.no_pipeline and headline
    if not no_pipeline:
        # T-PPR query of batch b+1 on a side stream, beside aggregate/update of batch b.  Several ranks: the same masks; the
        # row exchange (RCCL) is enqueued by the library on the CU-masked main stream, so it never lands on the T-PPR
        # stream's compute units.  ZT_BENCH_NO_MASKS=1 switches the masks off.
        if os.environ.get("ZT_BENCH_NO_MASKS") == "1":
            tppr_cus = 0
        try:
            tgn.enable_pipeline(tppr_cus=tppr_cus, group=group)
        except Exception as exc:                       # no CU-mask support: plain streams
            sys.stderr.write("[bench] CU-masked streams unavailable (%s); using plain streams\n" % exc)
            tppr_cus = 0
            tgn.enable_pipeline(tppr_cus=0, group=group)
    else:
        tppr_cus = 0
    main_stream = getattr(tgn, "main_stream", None)
    # several ranks (or --exchange-world1): every step of the native pipeline ends with the row exchange, enqueued by the
    # library itself (csrc/exchange.hip: pack -> ncclAllGather -> scatter -> projected-row refresh on the main stream); ranks
    # rehearsing on ONE GPU cannot form an RCCL communicator and go through shared memory instead
    xchg = None
    if not no_pipeline and (world > 1 or a.exchange_world1):
        xchg = "shm" if (rehearsal and world > 1) else "rccl"
        tgn.enable_exchange(rank, world, transport=xchg, shm_name="zt_bench_%s_%s" % (os.environ.get("MASTER_PORT", "0"), name))

    # views of every batch, made once: slicing tensors is host work that is not part of the path
    batches = [(src_d[b * bs:(b + 1) * bs], dst_d[b * bs:(b + 1) * bs], neg_d[b * bs:(b + 1) * bs],
                ts_d[b * bs:(b + 1) * bs], eidx_d[b * bs:(b + 1) * bs]) for b in range(n_steps_total)]
    state = dict(look=synth.pipeline_look(group))   # batches in sight: the rest of this group, the next group, the one after
                                                    # it, and one more (a group is only full while a follower is in sight)

    # the batch loop itself: native (zt_pipeline_run: the region's steps from one host call -- the loop of
    # evaluation/evaluation.py:19-45 in the library, the row exchange of a multi-rank run included) unless --python-loop or
    # no pipeline
    native = not no_pipeline and not a.python_loop
    prepared = {}

    def prep(b0, nb):
        if native:
            prepared[(b0, nb)] = tgn.prepare_run(batches[b0:b0 + nb])

    def run(b0, nb, keep=None):
        # exactly nb steps; nothing of step b0+nb is enqueued (the view ahead ends with the region).  keep: a [nb, 3 bs, H]
        # tensor that receives every step's embeddings (parity_in_run) instead of one buffer overwritten by every step
        ctx = torch.cuda.stream(main_stream) if main_stream is not None else contextlib.nullcontext()
        with ctx:
            if native:
                if (b0, nb) not in prepared:
                    prep(b0, nb)
                tgn.run_device(prepared[(b0, nb)], out=keep, look=state["look"])
                return
            for b in range(b0, b0 + nb):
                ahead = [] if no_pipeline else batches[b + 1: min(b + 1 + state["look"], b0 + nb)]
                e = step(*batches[b], ahead=ahead)
                if keep is not None:
                    keep[b - b0].copy_(e)

    lib = _capi.lib()
    if os.environ.get("ZT_DUMP_MAPS") and rank == 0:
        # (diagnostics: the process's memory map, so that the raw addresses of a native stack trace -- a profiler crash --
        #  can be put to library + offset afterwards: tools/symbolise.py)
        with open("/proc/self/maps") as fi, open(os.environ["ZT_DUMP_MAPS"], "w") as fo:
            fo.write(fi.read())
    _capi.set_kernel_choice(_capi.CHOICE_TPPR_PREPASS, _capi.PREPASS_COOP if a.prepass_coop else 0)
    _capi.set_kernel_choice(_capi.CHOICE_TPPR_CHAIN, _capi.CHAIN_PAIRED if a.chain_pairs else (a.chain_mode or 0))
    run(0, prefill)
    tgn.embedding_module.tppr_finder.check_status() if wl["strategy"] == "streaming" else None
    torch.cuda.synchronize()
    # ---- warm state for the CPU leg (the state after the 10 % prefill, where the GPU leg's warm-up steps start)
    # and row-fill statistics.  Taken BEFORE the warm-up so that the W warm-up steps run right before the timed
    # region (the snapshot idles the GPU for seconds).
    e0 = prefill * bs
    cpu_nb = 0
    if world == 1 and cpu_edges != 0:
        cpu_nb = max(2, min(steps + warmup, cpu_edges // bs))
    snap, fill = None, None
    if cpu_nb:
        touched = np.unique(np.concatenate([src[:e0], dst[:e0]])) if e0 else np.zeros(0, np.int64)
        snap = snapshot_state(tgn, wl, touched)
        if wl["strategy"] == "streaming" and len(touched):
            ends = np.concatenate([src[e0:e0 + bs], dst[e0:e0 + bs], neg[e0:e0 + bs]])
            pos = np.searchsorted(touched, ends)
            seen = (pos < len(touched)) & (touched[np.minimum(pos, len(touched) - 1)] == ends)
            ln = np.where(seen, snap["tppr"][0]["len"][np.minimum(pos, len(touched) - 1)], 0)
            fill = dict(mean_row_len=float(ln.mean()), frac_empty=float((ln == 0).mean()), frac_full=float((ln == k).mean()))
    spun = False
    if cpu_nb and not a.no_clock_spin:
        # the snapshot above left the GPU idle for seconds (host-side export of the warm state): its clocks have dropped,
        # and W = 5 warm-up steps (2 ms) do not bring them back.  Throw-away matrix products for a third of a second, on
        # data that has nothing to do with the stream, before the warm-up steps -- so that the timed region measures the
        # path and not the clock ramp
        xa = torch.randn((4096, 4096), device=device)
        t_spin = time.perf_counter()
        while time.perf_counter() - t_spin < 0.35:
            for _ in range(20):
                xa = torch.mm(xa, xa).clamp_(-1.0, 1.0)
            torch.cuda.synchronize()
        del xa
        spun = True
    prep(prefill + warmup, steps)               # (the region's batch list in the library's form: made before the clock starts)
    # parity_in_run: the embeddings of the warm-up and the timed steps are KEPT (one [3 bs, H] block per step instead of one
    # buffer every step overwrites) and compared with the oracle's after the timed regions, in cpu_baseline()
    keep_w = keep_t = None
    if cpu_nb:
        Hd = tgn.embedding_dimension * (M + 1)
        keep_w = torch.empty((warmup, 3 * bs, Hd), dtype=torch.float32, device=device) if warmup else None
        keep_t = torch.empty((steps, 3 * bs, Hd), dtype=torch.float32, device=device)
    run(prefill, warmup, keep_w)
    if not a.no_profile:
        lib.zt_profile_reset()
        # (two event records per timed launch are host calls too: every 4th launch of the main stream's kernels in the headline
        #  run, every 8th in the legs, whose small-batch steps are bound by host enqueue)
        lib.zt_profile_enable(a.profile_every if a.profile_every > 0 else (4 if headline else 8))
    if wl["strategy"] == "streaming":
        tgn.embedding_module.tppr_finder.chain_stats()          # (clears the counters: what follows is the timed region's)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(prefill + warmup, steps, keep_t)
    t_host = time.perf_counter() - t0          # host time to enqueue the timed steps
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    sys.stderr.write("[bench] %s: host enqueue %.3f ms/step, wall %.3f ms/step\n" % (name, 1e3 * t_host / steps, 1e3 * dt / steps))
    chain_stats = tgn.embedding_module.tppr_finder.chain_stats() if wl["strategy"] == "streaming" else None
    lib.zt_profile_enable(0)
    # parity_in_run: the state right behind the timed region (T-PPR rows, memory rows of every node the warm-up and timed
    # batches touched), when the CPU sample ends exactly there
    after = None
    if cpu_nb == warmup + steps:
        ea, eb = prefill * bs, (prefill + cpu_nb) * bs
        ids_a = np.unique(np.concatenate([src[ea:eb], dst[ea:eb], neg[ea:eb]])).astype(np.int64)
        after = snapshot_state(tgn, wl, ids_a)
        if spun:                                   # (the export idled the GPU again: same throw-away work before the next region)
            xa = torch.randn((4096, 4096), device=device)
            t_spin = time.perf_counter()
            while time.perf_counter() - t_spin < 0.35:
                for _ in range(20):
                    xa = torch.mm(xa, xa).clamp_(-1.0, 1.0)
                torch.cuda.synchronize()
            del xa
    # ---- the second region: the same step + the link scorer (sharded runs score nothing: a rank holds a row shard) ----
    with_scorer = None
    if scored:
        kern_a = {}
        if not a.no_profile:
            for kn in ("tppr_prepass", "tppr_stream", "tppr_cleanup", "pruned_topk", "embed_prep", "fc1_agg", "embed_out",
                       "store_messages", "gru_update", "exchange"):
                n, ms = C.c_int64(), C.c_double()
                lib.zt_profile_read(kn.encode(), C.byref(n), C.byref(ms))
                if n.value:
                    kern_a[kn] = dict(launches=n.value, avg_us=1e3 * ms.value / n.value)
        tgn.enable_scoring()
        b1 = prefill + warmup + steps
        prep(b1 + 5, steps)
        run(b1, 5)
        if not a.no_profile:
            lib.zt_profile_reset()
            lib.zt_profile_enable(a.profile_every if a.profile_every > 0 else 4)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run(b1 + 5, steps)
        torch.cuda.synchronize()
        dt_s = time.perf_counter() - t1
        lib.zt_profile_enable(0)
        n, ms = C.c_int64(), C.c_double()
        lib.zt_profile_read(b"score", C.byref(n), C.byref(ms))
        with_scorer = dict(value=steps * bs / dt_s, unit="edges/s", ms_per_step=1e3 * dt_s / steps, steps=steps,
                           score_kernel_us=(1e3 * ms.value / n.value) if n.value else None,
                           note="the step above + compute_edge_probabilities' scorer (zt_affinity) at its tail, timed over the "
                                "next %d batches after 5 warm-up steps; `value` is the metric's step (embeddings + T-PPR and "
                                "memory updates)" % steps)
        sys.stderr.write("[bench] %s: with the scorer %.3f ms/step\n" % (name, 1e3 * dt_s / steps))
    # ---- the third region (headline): the metric's step over `steady` batches, launch groups as such a region gets them ----
    steady_state = None
    if steady:
        if scored:
            tgn.enable_scoring(False)
        _, g_s = synth.pipeline_settings(wl, steady, -1, a.group if headline else -1)
        if g_s != group and wl["strategy"] == "streaming":
            _capi.check(lib.zt_pipeline_set_group(tgn._pipe, C.c_int32(g_s)), "zt_pipeline_set_group")
            tgn._pipe_group = g_s
        state["look"] = synth.pipeline_look(g_s)
        b2 = prefill + warmup + steps + ((5 + steps) if scored else 0)
        prep(b2 + 10, steady)
        run(b2, 10)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        run(b2 + 10, steady)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt_ss = time.perf_counter() - t2
        if world > 1:
            tm2 = torch.tensor([dt_ss], dtype=torch.float64, device=device)
            dist.all_reduce(tm2, op=dist.ReduceOp.MAX)
            dt_ss = float(tm2.item())
        steady_state = dict(value=steady * bs / dt_ss, unit="edges/s", ms_per_step=1e3 * dt_ss / steady, steps=steady, warmup=10,
                            tppr_launch_group=g_s,
                            note="the metric's step (same workload, same settings, no per-kernel events) over %d batches behind the "
                                 "regions above: the timed region's first batch (prepass + a single-batch T-PPR update before any "
                                 "aggregation can start) and its last aggregation weigh 1/%d here instead of 1/%d" % (steady, steady, steps))
        sys.stderr.write("[bench] %s: steady state %.3f ms/step over %d steps\n" % (name, 1e3 * dt_ss / steady, steady))
    if wl["strategy"] == "streaming":
        tgn.embedding_module.tppr_finder.check_status()
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    torch.cuda.synchronize()
    use_proj = bool(getattr(tgn.embedding_module, "use_projection", False))
    weights = time_w = efeat_host = None
    if cpu_nb:
        weights, time_w = model_weights(tgn)
        if F != 1:
            efeat_host = tgn.edge_raw_features.cpu().numpy()
    if not no_pipeline:
        tgn.enable_pipeline(False)      # give the CU-masked streams back before the runtime shuts down (or the next workload)

    # ---- per-kernel HIP-event times over the timed region ----
    kern = {}
    if scored:
        kern = kern_a
    elif not a.no_profile:
        for kn in ("tppr_prepass", "tppr_stream", "tppr_cleanup", "pruned_topk", "embed_prep", "fc1_agg",
                   "embed_out", "store_messages", "gru_update", "exchange"):
            n, ms = C.c_int64(), C.c_double()
            lib.zt_profile_read(kn.encode(), C.byref(n), C.byref(ms))
            if n.value:
                kern[kn] = dict(launches=n.value, avg_us=1e3 * ms.value / n.value)
    if "gru_update" in kern and "embed_out" not in kern and "fc1_agg" in kern:
        # (k_out_gru / k_out_gru2: the output layers run inside the GRU update's launch)
        kern["gru_update"]["note"] = "one launch with the output layers (k_out_gru): embed_out has no entry of its own"
    cus_total = torch.cuda.get_device_properties(device).multi_processor_count
    # the model's tables are the bulk of the device memory: gone before the next workload (or the CPU leg's host arrays)
    del tgn, batches, src_d, dst_d, neg_d, ts_d, eidx_d
    torch.cuda.empty_cache()
    if rank != 0:
        return None

    edges = steps * bs
    # HBM bytes per launch from the committed rocprofv3 PMC passes of this workload (separate
    # FETCH_SIZE / WRITE_SIZE runs, gfx950 correction; see profiles/make_pmc_summary.py)
    pmc = {}
    pmc_edges = None
    pmc_source = None
    pmc_stale = None
    if world == 1:
        for rnd in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True):
            pj = os.path.join(ROOT, "profiles", rnd, "%s_pmc_summary.json" % name)
            if os.path.isfile(pj):
                summ = json.load(open(pj))
                pmc_source = "profiles/%s/%s_pmc_summary.json (commit %s)" % (rnd, name, (summ.get("measured_on") or {}).get("commit"))
                pmc = summ.get("kernels", {})
                # per-launch bytes of the T-PPR kernel belong to a launch SHAPE: a summary measured on launches of another
                # size is scaled by the edges per launch (the kernel's traffic is per edge -- rows read and written, tag
                # polls), and the roofline note says so
                on = summ.get("measured_on") or {}
                # the counters belong to a LIBRARY: a summary taken on other kernel sources than the ones this run was built
                # from is quoted all the same (the launch shapes and byte counts move little) but flagged
                try:
                    from zebra_amd.build import csrc_sha16
                    pmc_stale = on.get("csrc_sha16") != csrc_sha16()
                except Exception:
                    pmc_stale = None
                pmc_edges = on.get("edges_per_k_stream_launch")
                if on.get("tppr_launch_group") not in (None, group) and not pmc_edges:
                    pmc = {kk: vv for kk, vv in pmc.items() if kk != "tppr_stream"}
                break
    ab = algorithmic_bytes(k, F, M, strategy=wl["strategy"], width=wl.get("width", 10), depth=wl.get("depth", 2))
    af = algorithmic_flops(k, F, M)
    ex = executed_flops(k, F, M, projected=use_proj)

    def kernel_roofline(kn):
        """achieved = ALGORITHMIC bytes (or EXECUTED flops) of one launch / its average HIP-event time."""
        us = kern[kn]["avg_us"] * 1e-6
        shard = world if kn in ("fc1_agg", "embed_out", "pruned_topk") and world > 1 else 1
        tr = pmc.get(kn, {}).get("traffic")
        if kn in ("fc1_agg", "embed_out"):
            ach = ex[kn] * bs / shard / us / 1e12
            # the main stream is confined to the CUs the T-PPR stream does not own (CU masks): the whole-chip peak is
            # the contract's `peak`; the share of the CUs the kernel can run on is given beside it
            cus_used = cus_total - (tppr_cus if (not no_pipeline and tppr_cus > 0) else 0)
            return dict(kernel=kn, bound="mfma", achieved=ach, peak=MFMA_F32_PEAK_TF, unit="TFLOP/s",
                        frac=ach / MFMA_F32_PEAK_TF, traffic=tr, traffic_source=pmc_source if tr is not None else None, cus=cus_used,
                        frac_of_own_cus=ach / (MFMA_F32_PEAK_TF * cus_used / cus_total),
                        note="FLOPs the kernel executes (fc2 runs after the k-reduction, in embed_out; W_m memory[v] "
                             "comes from the projected table); the reference formulation would count %.2fx more" % (M * af["p2"] / ex["fc1_agg"]) if kn == "fc1_agg" else None)
        if kn == "tppr_stream":
            # one k_stream launch covers all M models and `group` batches (fewer at the ends of the region)
            per_launch = steps * bs / kern[kn]["launches"]     # (every launch of the T-PPR update is timed)
            byts = ab["p1"] * M * per_launch
            note = ("dependency/latency-bound phase (SURVEY.md 8d P1): edges of a batch are applied in order along "
                    "per-node chains; the binding resource is hops x hop latency, not HBM")
            if tr is not None and pmc_edges and abs(pmc_edges - per_launch) > 1:
                tr = tr * per_launch / pmc_edges
                note += "; traffic: counters of %d-edge launches scaled to this run's %.0f edges per launch" % (pmc_edges, per_launch)
        elif kn == "pruned_topk":
            # ONE launch serves all M models: the CSR tails are read once, every model writes its output rows
            byts = (ab["p1"] - 48 * k + M * 48 * k) * bs / shard
            note = ("CSR tail reads (once for all models) + output rows per query and model (search probes not counted); the kernel is "
                    "ONE wavefront per query row and the whole launch is resident at once (3 B waves on the chip), so its time is a "
                    "wave's chain of DEPENDENT memory round trips + its selection: latency_model")
        elif kn == "gru_update":
            byts = 2 * (4 * (2 * 100 + F + 100) + 8 * 100 + 4) * bs / shard
            note = None
        else:
            byts = ab["p3"] * bs / shard
            note = None
        ach = byts / us / 1e9
        # the streaming T-PPR kernel is bound by (hops of the longest per-node chain) x (latency of one hop), not by
        # HBM: `achieved` / `peak` / `frac` are still its algorithmic bytes against the HBM peak, for the record
        out_r = dict(kernel=kn, bound="latency" if kn == "tppr_stream" else "hbm", achieved=ach, peak=HBM_PEAK_GBS,
                     unit="GB/s", frac=ach / HBM_PEAK_GBS, traffic=tr, traffic_source=pmc_source if tr is not None else None, note=note)
        if kn == "pruned_topk":
            # a query row's dependent chain (csrc/tppr_prune.hip): per level of the walk indptr -> the P-ary search of every
            # frontier entry (P = 64 lanes for the single entry of level 0, 8 per entry for a full level: ceil(log_P(degree + 1))
            # rounds of probes, each a round trip) -> the tails (one round of coalesced loads); degrees from the run's own
            # adjacency: level 0 the batch's query nodes, level 1 their neighbours (degree-weighted: a neighbour is drawn by
            # an edge).  One round trip = an Infinity-Cache hit (227 ns, MI355X_MICROARCH.md: the CSR of C4 is 48 MB -- beyond
            # the 4 MB L2 of an XCD, inside the 256 MB cache).  The selection (de-duplication in the reference's summation
            # order + the replayed argsort: compute, 26 us alone -- tools/exp/prune_phases.py) is given beside the chain.
            tb0 = prefill + warmup
            deg = np.bincount(np.concatenate([src, dst]), minlength=wl["n_nodes"] + 2).astype(np.float64)
            qn = np.concatenate([src[tb0 * bs:(tb0 + steps) * bs], dst[tb0 * bs:(tb0 + steps) * bs], neg[tb0 * bs:(tb0 + steps) * bs]])
            lg = lambda d_, P: np.ceil(np.log(np.maximum(d_, 1.0) + 1.0) / np.log(P))
            r0 = float(lg(deg[qn], 64.0).max())                       # (the launch ends with its slowest wave: the largest degree)
            r0_mean = float(lg(deg[qn], 64.0).mean())
            dn = deg[deg > 0]
            r1 = float(lg(dn, 8.0).max())
            r1_mean = float((lg(dn, 8.0) * dn).sum() / dn.sum())
            depth_ = wl.get("depth", 2)
            trips_mean = (1 + r0_mean + 1) + (depth_ - 1) * (1 + r1_mean + 1)
            trips_max = (1 + r0 + 1) + (depth_ - 1) * (1 + r1 + 1)
            rt_ns, sel_us = 227.0, 26.0
            kt_us = kern[kn]["avg_us"]
            out_r["latency_model"] = dict(
                dependent_round_trips_mean=trips_mean, dependent_round_trips_slowest_wave=trips_max, ns_per_round_trip=rt_ns,
                memory_chain_floor_us=trips_max * rt_ns / 1e3, selection_us_alone=sel_us,
                floor_us=trips_max * rt_ns / 1e3 + sel_us, kernel_us=kt_us, frac=(trips_max * rt_ns / 1e3 + sel_us) / kt_us,
                note="floor = (dependent round trips of the slowest wave: per level indptr + ceil(log_P(degree + 1)) search rounds + the "
                     "tails) x an Infinity-Cache hit + the selection phase measured alone; frac = floor / kernel time")
        if kn == "tppr_stream":
            out_r["edges_per_launch"] = per_launch
            # the yardstick that fits: the longest chain of edges through ONE node in a launch (every edge that touches
            # the node is a hop of its chain: they are applied one after the other) x the chain's critical section
            tb0 = prefill + warmup
            hops = []
            for g0 in range(tb0, tb0 + steps, group):
                g1 = min(g0 + group, tb0 + steps)
                ends = np.concatenate([src[g0 * bs:g1 * bs], dst[g0 * bs:g1 * bs][src[g0 * bs:g1 * bs] != dst[g0 * bs:g1 * bs]]])
                hops.append(int(np.bincount(ends).max()))
            ch = float(np.mean(hops))
            kt_ns = kern[kn]["avg_us"] * 1e3
            out_r["latency_model"] = dict(
                critical_hops=ch, ns_per_hop=kt_ns / ch, floor_ns_per_hop=CRIT_SECTION_NS, frac=ch * CRIT_SECTION_NS / kt_ns,
                note="critical_hops: edges through the most-touched node of a launch (mean over the region's nominal launches "
                     "of %d batches), applied in stream order; floor: the hub chain's critical section (2280 core clocks, median, "
                     "tools/crit_profile.py); frac = hops x floor / kernel time" % group)
        return out_r

    # dominant kernel = the one with the largest total time in the timed region
    roof = roof2 = None
    if kern:
        dom = max(kern, key=lambda n: kern[n]["avg_us"] * kern[n]["launches"])
        roof = kernel_roofline(dom)
        # the largest THROUGHPUT kernel as well, when the dominant one is the latency-bound T-PPR chain
        if dom != "fc1_agg" and "fc1_agg" in kern:
            roof2 = kernel_roofline("fc1_agg")
        for r_ in (roof, roof2):
            if r_ is not None and r_.get("traffic") is not None:
                r_["traffic_stale"] = pmc_stale       # True: the summary's kernel sources are not this run's (csrc_sha16)

    cpu = None
    if cpu_nb:
        b0 = prefill
        cb = [(src[b * bs:(b + 1) * bs], dst[b * bs:(b + 1) * bs], neg[b * bs:(b + 1) * bs], ts[b * bs:(b + 1) * bs],
               eidx[b * bs:(b + 1) * bs]) for b in range(b0, b0 + cpu_nb)]
        extra = dict(src=src, dst=dst, eidx=eidx, ts=ts)
        if F != 1:
            extra["efeat"] = efeat_host
        def gpu_emb(b, kw=keep_w, kt=keep_t):
            t = kw[b] if b < warmup else (kt[b - warmup] if b - warmup < steps else None)
            return None if t is None else t.cpu().numpy()
        cpu = cpu_baseline(wl, snap, weights, time_w, cb, extra, min(16, os.cpu_count() or 1),
                           gpu=dict(emb=gpu_emb, warmup=warmup, after=after))
        del keep_w, keep_t

    value = edges / dt
    out = {
        "metric": "temporal edges/sec embedded (k=%d, 2-layer)" % k if M == 2 else
                  "temporal edges/sec embedded (k=%d, %d T-PPR models)" % (k, M),
        "value": value, "unit": "edges/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": 1e3 * dt / steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64+f32", "data": "synthetic",
        "config": {"workload": "%s: synthetic %s stream, %d nodes, bs=%d, k=%d, alpha=%s beta=%s, %s T-PPR, F=%d, "
                               "prefill %d + warmup %d batches" % (name, "bipartite" if wl["bipartite"] else
                                                                   "power-law", wl["n_nodes"], bs, k, wl["alpha"],
                                                                   wl["beta"], wl["strategy"], F, prefill, warmup),
                   "global_batch": bs, "tppr_launch_group": group, "tppr_cus": tppr_cus,
                   "tppr_group_release": "launch (event)" if (a.release_by_launch or a.release_by_launch_full) else "member (a counter per batch inside the launch)",
                   "clock_spin": spun, "step_loop": "native (zt_pipeline_run)" if native else "python",
                   "row_exchange": None if xchg is None else ("in the native step: %s, %d rank%s, [id | memory row | last_update] per touched row"
                                                               % ("RCCL ncclAllGather" if xchg == "rccl" else "shared memory (ranks share one GPU)",
                                                                  world, "" if world == 1 else "s")),
                   "node_ids": "id == popularity rank" if a.perm_seed < 0 else "shuffled (perm_seed %d)" % a.perm_seed,
                   "parallelism": "replicated T-PPR + row-sharded aggregate x%d" % world},
        "host_enqueue_ms_per_step": 1e3 * t_host / steps,
        "exchange_us_per_step": kern["exchange"]["avg_us"] if "exchange" in kern else None,
        # the replicated part of a multi-GPU step: every rank replays the whole batch's T-PPR chain (streaming strategy), so
        # this much of the step does not shrink with N -- a flat curve is explained by the line itself
        # hub chains of the timed region, all chains and models: positions that shared a critical section with their neighbour
        # (csrc/tppr_pair.hpp) against positions taken singly -- the hit rate of the paired hop, measured in the kernel
        "chain_hops": None if (chain_stats is None or not (a.chain_pairs or a.chain_mode)) else dict(
            chain_stats, paired_share=(2.0 * chain_stats["pairs_done"] /
                                       max(1, 2 * chain_stats["pairs_claimed"] + chain_stats["singles"]))),
        "chain_bound_ms_per_step": (kern["tppr_stream"]["avg_us"] * kern["tppr_stream"]["launches"] / steps / 1e3)
                                   if "tppr_stream" in kern else None,
        "with_scorer": with_scorer,
        "steady_state": steady_state,
        "parity_in_run": (cpu or {}).pop("parity_in_run", None),
        "roofline": roof,
        "roofline_throughput_kernel": roof2,
        "cpu_baseline": cpu,
        "row_fill": fill,
        # flops_per_edge: SURVEY.md 8(d)'s figure = the REFERENCE's formulation (fc2 per neighbour, W_m memory per
        # gathered row, scorer included: `value`'s step ends with the embeddings, `with_scorer` times the scorer too);
        # executed_flops_per_edge: what the kernels of `value`'s step execute; the scorer kernel adds
        # scorer_executed_flops_per_edge (W_a src shared by an edge's two pairs: 3/4 of the reference's)
        "algorithmic": {"bytes_per_edge": ab["total"], "flops_per_edge": af["total"],
                        "executed_flops_per_edge": ex["fc1_agg"] + ex["embed_out"] + ex["project_rows"] + af["gru"],
                        "scorer_flops_per_edge": af["scorer"], "scorer_executed_flops_per_edge": af["scorer"] * 3 // 4,
                        "hbm_gbs_at_value": ab["total"] * value / 1e9,
                        "hbm_frac_at_value": ab["total"] * value / 1e9 / HBM_PEAK_GBS,
                        "reference_formulation_tflops_at_value": af["total"] * value / 1e12,
                        "reference_formulation_mfma_frac_at_value": af["total"] * value / 1e12 / MFMA_F32_PEAK_TF},
        "kernels": kern,
    }
    out["algorithmic"]["executed_mfma_frac_at_value"] = out["algorithmic"]["executed_flops_per_edge"] * value / 1e12 / MFMA_F32_PEAK_TF
    return out


def run_train_workload(a, name, steps, warmup, device):
    """The TRAINING step of workload `name` (streaming strategy): train.py:195-215 -- compute_edge_probabilities(train=True)
    through the drop-in's reference surface (numpy batches in, as train.py hands them over), BCE on the positive and negative
    probabilities, backward through the fused HIP kernels (aggregate_bwd.hip, train_ops.hip), Adam step, dropout on -- timed
    over `steps` batches after `warmup`, from a state the first 10 % of the stream has filled (eval steps through the native
    pipeline: rows full, memory warm).  Beside it the same step on the host: oracle/torch_cpu_train.py (the torch-CPU
    restatement pinned to the reference's own gradients, T-PPR by the C port on one thread), started from the same warm
    state.  The one figure of this bench that can be held against a PUBLISHED one."""
    import torch
    from zebra_amd import synth

    base = name[:-len("_train")]
    wl = dict(synth.WORKLOADS[base])
    bs, k, M, F = wl["bs"], wl["k"], len(wl["alpha"]), wl["F"]
    prefill = (wl["n_edges"] // 10) // bs
    n_total = prefill + warmup + steps
    src, dst, neg, ts, eidx = make_stream(wl, n_total * bs, perm_seed=None if a.perm_seed < 0 else a.perm_seed)
    n_edge_rows = (wl["n_edges"] if F == 1 else n_total * bs) + 1
    tgn = build_model(wl, device, n_edge_rows)
    drop_p = float(tgn.embedding_module.drop.p)
    # ---- warm state: the first 10 % of the stream as eval steps (native loop), then the pipeline goes ----
    if prefill:
        d = [torch.from_numpy(x[:prefill * bs]).to(device) for x in (src, dst, neg, ts, eidx)]
        tppr_cus, group = synth.pipeline_settings(wl, prefill)
        tgn.enable_pipeline(tppr_cus=tppr_cus, group=group)
        bt = [tuple(x[b * bs:(b + 1) * bs] for x in d) for b in range(prefill)]
        with torch.cuda.stream(tgn.main_stream):
            tgn.run_device(tgn.prepare_run(bt), look=synth.pipeline_look(group))
        torch.cuda.synchronize()
        tgn.embedding_module.tppr_finder.check_status()
        tgn.enable_pipeline(False)
        del d, bt
    e0 = prefill * bs
    touched = np.unique(np.concatenate([src[:e0], dst[:e0]])) if e0 else np.zeros(0, np.int64)
    snap = snapshot_state(tgn, wl, touched)
    weights, time_w = model_weights(tgn)
    efeat_host = tgn.edge_raw_features.cpu().numpy()
    tgn.train()
    opt = torch.optim.Adam(tgn.parameters(), lr=1e-4)                 # train.py:29,150
    crit = torch.nn.BCELoss()
    ones, zeros = torch.ones(bs, device=device), torch.zeros(bs, device=device)
    losses = []

    def step(b):
        s_, e_ = b * bs, (b + 1) * bs
        opt.zero_grad()
        pos, negp = tgn.compute_edge_probabilities(src[s_:e_], dst[s_:e_], neg[s_:e_], ts[s_:e_], eidx[s_:e_], 10, True)
        loss = crit(pos.squeeze(), ones) + crit(negp.squeeze(), zeros)
        loss.backward()
        opt.step()
        tgn.memory.detach_memory()
        return loss

    for b in range(prefill, prefill + warmup):
        step(b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in range(prefill + warmup, n_total):
        losses.append(step(b))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    loss_first, loss_last = float(losses[0].item()), float(losses[-1].item())
    tgn.embedding_module.tppr_finder.check_status()
    del tgn, opt
    torch.cuda.empty_cache()
    sys.stderr.write("[bench] %s: %.3f ms/step (training)\n" % (name, 1e3 * dt / steps))
    # ---- the same step on the host, from the same warm state ----
    cpu = None
    if a.cpu_edges != 0:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import torch_cpu_train
        n_thr = min(16, os.cpu_count() or 1)
        keep_thr = torch.get_num_threads()
        tr = torch_cpu_train.TorchCpuTrainer(wl["n_nodes"] + 1, 100, F, 100, k, wl["alpha"], wl["beta"], weights, efeat_host, time_w,
                                             dropout=drop_p, lr=1e-4, n_threads=n_thr)
        ids = snap["ids"]
        for m in range(M):
            tr.tppr.import_rows(m, ids, snap["tppr"][m])
        tr.mem.memory[ids] = snap["memory"]
        tr.mem.last_update[ids] = snap["last_update"]
        tr.mem.messages[ids] = snap["messages"]
        tr.mem.timestamps[ids] = snap["timestamps"]
        tr.mem.flags[ids] = snap["flags"]
        nb_cpu = max(3, min(steps + warmup, 12))
        t_c, n_c, l_c = 0.0, 0, []
        for q, b in enumerate(range(prefill, prefill + nb_cpu)):
            s_, e_ = b * bs, (b + 1) * bs
            t1 = time.perf_counter()
            l_c.append(tr.step(src[s_:e_], dst[s_:e_], neg[s_:e_], ts[s_:e_], eidx[s_:e_]))
            if q >= 2:                                   # (two steps to warm the allocator and the thread pool)
                t_c += time.perf_counter() - t1
                n_c += bs
        torch.set_num_threads(keep_thr)
        cpu = dict(value=n_c / t_c, unit="edges/s", cores=n_thr, kind="port", host_cpus=os.cpu_count(),
                   sample="%d training steps (%d edges timed, the first two steps untimed) of the same stream from the GPU run's warm "
                          "state: oracle/torch_cpu_train.py -- T-PPR by the C port on 1 thread, everything else torch-CPU ops with "
                          "autograd on %d threads, Adam step included" % (nb_cpu, n_c, n_thr),
                   p1_share=tr.t_tppr / max(t_c, 1e-9), loss_first_step=l_c[0])
    return {
        "value": steps * bs / dt, "unit": "edges/s", "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * dt / steps,
        "config": {"workload": "%s: the TRAINING step on %s's stream (%d nodes, bs=%d, k=%d, %d T-PPR models, F=%d): "
                               "compute_edge_probabilities(train=True) + BCE + backward + Adam(lr 1e-4), dropout %.1f, numpy batches "
                               "handed over per step as train.py does; prefill %d eval batches + warmup %d"
                               % (name, base, wl["n_nodes"], bs, k, M, F, drop_p, prefill, warmup),
                   "global_batch": bs, "step_loop": "python (the reference's training loop, train.py:195-215)"},
        "loss_first_timed_step": loss_first, "loss_last_timed_step": loss_last,
        "cpu_baseline": cpu,
        "published_context": {"value": 12000.0, "unit": "edges/s (derived upper bound)",
                              "source": "technical_report.pdf p.12 Table 6: Wikipedia, m = 2, k = 20: 8.91 s per training epoch on an RTX 2080 Ti + "
                                        "Xeon 2.60 GHz (Numba T-PPR on the host); <= 110 K training edges / 8.91 s (BASELINE.md section 1)",
                              "note": "other hardware, the real Wikipedia stream, an epoch that starts from empty state: context, not a "
                                      "same-node comparison; vs_baseline stays null"},
        "parity": "the training path is held to the reference's own loss and gradients by fixture g8_train_grads "
                  "(tests/test_embed_gpu.py::test_training_step_gradients_match_reference; the CPU restatement timed here by "
                  "tests/test_oracle_golden.py::test_torch_cpu_training_step_matches_reference_gradients)",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c5", choices=["c1", "c2", "c3", "c4", "c5"])
    ap.add_argument("--legs", default="default",
                    help="further workloads timed after the headline one and printed under \"workloads\" of the same JSON "
                         "line, each with its own ms_per_step / roofline / cpu_baseline: comma-separated names, 'none', or "
                         "'default' = c2,c3,c2_train,c4 behind the c5 headline (nothing behind another --workload); <name>_train = the TRAINING "
                         "step of that workload (run_train_workload)")
    ap.add_argument("--leg-steps", type=int, default=100, help="timed steps of every leg (warm-up 10)")
    ap.add_argument("--steady-steps", type=int, default=200,
                    help="a further timed region of this many steps behind the headline's (printed as \"steady_state\"; skipped "
                         "when --steps is at least as long; 0 = none)")
    ap.add_argument("--prefill-steps", type=int, default=-1,
                    help="untimed batches run before warm-up so that T-PPR rows are full (default: per workload)")
    ap.add_argument("--cpu-edges", type=int, default=-1, help="edges of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-profile", action="store_true", help="do not record per-kernel HIP events")
    ap.add_argument("--no-score", action="store_true", help="skip the second timed region (the step with the link scorer at its tail)")
    ap.add_argument("--no-clock-spin", action="store_true",
                    help="skip the throw-away GPU work between the warm-state snapshot (seconds of host work, GPU idle) and the warm-up steps")
    ap.add_argument("--profile-every", type=int, default=-1,
                    help="record HIP events around every n-th launch of the main stream's kernels (default 4: two "
                         "event records per kernel and step cost ~3 %% of the step; the T-PPR update, one launch per "
                         "group of batches on its own stream, is timed at every launch)")
    ap.add_argument("--tppr-cus", type=int, default=-1,
                    help="pin the T-PPR stream to this many compute units (CU mask) and everything else to the rest "
                         "(0 = no masks; default: two whole XCDs = 64 for the streaming strategy, "
                         "0 for the pruning strategy, whose query kernel wants the whole chip)")
    ap.add_argument("--group", type=int, default=-1,
                    help="consecutive batches whose streaming T-PPR update runs as ONE launch (zt_pipeline_set_group); "
                         "default: as many as fit a launch (<= 16384 edges), at most 4; 1 for the pruning strategy")
    ap.add_argument("--python-loop", action="store_true",
                    help="one Python call per step (TGN.step_device) instead of the library's batch loop (TGN.run_device)")
    ap.add_argument("--chain-mode", type=int, default=0,
                    help="hub chains of the T-PPR update: 0 the library's pick, 1 single hops through the mailbox, 3 spine (zt_set_kernel_choice)")
    ap.add_argument("--choice", action="append", default=[], metavar="SELECTOR=VALUE",
                    help="zt_set_kernel_choice(SELECTOR, VALUE) before the run (A/B of kernels that compute the same function; "
                         "numbers as in include/zebra_amd.h), repeatable")
    ap.add_argument("--release-by-launch", action="store_true",
                    help="the aggregation of a batch waits for the END of the T-PPR launch its group shares (the form before round 6, "
                         "with tapering groups) instead of for that batch's rows (A/B)")
    ap.add_argument("--release-by-launch-full", action="store_true",
                    help="release by launch in full, untapered groups: uniform T-PPR launches for counter passes (rocprofv3 --pmc "
                         "serialises kernels, so the release by member cannot run under it)")
    ap.add_argument("--prepass-coop", action="store_true", help="the dependency prepass of big launches as ONE cooperative kernel (A/B against the eleven launches)")
    ap.add_argument("--chain-pairs", action="store_true",
                    help="hub chains take TWO positions per critical section where they can (csrc/tppr_pair.hpp; zt_set_kernel_choice: "
                         "bit-exact, measured slower -- DESIGN.md section 5 --, off by default)")
    ap.add_argument("--exchange-world1", action="store_true",
                    help="one rank WITH the row exchange of a multi-GPU run in its step loop (a world-1 RCCL communicator made by the "
                         "library): what a one-GPU box can show of the N > 1 step -- host enqueue and kernel time of pack / all-gather "
                         "/ scatter / refresh")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="run the T-PPR query on the main stream instead of overlapping it with the previous batch")
    ap.add_argument("--perm-seed", type=int, default=7,
                    help="shuffle which node id carries which popularity rank (hot rows of the per-node tables are then "
                         "scattered, not contiguous); -1 = id == rank as in rounds 1-2")
    ap.add_argument("--dry-run", action="store_true",
                    help="rendezvous only (gloo, no GPU): checks that --gpus N really starts N ranks")
    a = ap.parse_args()

    # ---- --gpus N without a launcher: this process becomes the parent of N ranks and never touches the GPU ----
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (a.gpus, world))

    import torch
    import torch.distributed as dist

    if a.dry_run:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        n_seen = 1
        if world > 1:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            t = torch.ones(1)
            dist.all_reduce(t)
            n_seen = int(t.item())
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"metric": "dry run (rendezvous only)", "value": None, "n_gpus": world, "ranks_seen": n_seen,
                              "dry_run": True, "launched_by": "bench.py" if os.environ.get("ZT_BENCH_LAUNCHED") else "launcher"}))
        return

    # rehearsal on a one-GPU box: ZT_BENCH_REHEARSAL=1 puts every rank on cuda:0 and uses gloo
    rehearsal = os.environ.get("ZT_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
        # (several ranks on ONE GPU: every rank's T-PPR launches take 1 / world of the stream's CUs and run without hub
        #  chains -- run_workload: tppr_finder.set_device_share)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("gloo" if rehearsal else "nccl", rank=rank, world_size=world)
    assert torch.cuda.is_available(), "bench.py needs an MI355X; there is no CPU fallback"
    device = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(device)

    cpu_default = {"c5": 48 * 4096, "c3": 40 * 600, "c2": 60 * 200, "c1": 60 * 200, "c4": 20 * 1000}
    ce = a.cpu_edges if a.cpu_edges >= 0 else cpu_default[a.workload]
    out = run_workload(a, a.workload, a.steps, a.warmup, world, rank, device, True, ce, steady=a.steady_steps)
    legs = a.legs
    if legs == "default":
        legs = "c2,c3,c2_train,c4" if a.workload == "c5" else "none"
    res = {}
    for name in [x for x in legs.split(",") if x and x != "none"]:
        if name.endswith("_train"):
            if world == 1:
                r = run_train_workload(a, name, a.leg_steps, 10, device)
                if rank == 0:
                    res[name] = r
            continue
        # the other BASELINE configs, short: the driver's one run times all four (CPU legs bounded to ~2-4 s each)
        # (10 warm-up batches come first in the sample: 16 / 14 / 14 batches leave 6 / 4 / 4 TIMED ones for parity_in_run)
        leg_cpu = 0 if a.cpu_edges == 0 else {"c1": 16 * 200, "c2": 16 * 200, "c3": 14 * 600, "c4": 14 * 1000, "c5": 14 * 4096}[name]
        r = run_workload(a, name, a.leg_steps, 10, world, rank, device, False, leg_cpu)
        if r is not None:
            res[name] = {kk: r[kk] for kk in ("value", "unit", "steps", "warmup", "ms_per_step", "config", "host_enqueue_ms_per_step", "exchange_us_per_step", "chain_hops", "chain_bound_ms_per_step", "with_scorer", "parity_in_run",
                                              "roofline", "roofline_throughput_kernel", "cpu_baseline", "algorithmic", "kernels")}
    if rank == 0:
        out["rccl_ranks"] = world if (world > 1 and not rehearsal) else 0
        if res:
            out["workloads"] = res
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
