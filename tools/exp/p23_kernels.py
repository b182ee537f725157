#!/usr/bin/env python3
"""The organisations of the output layers (zt_embed's second kernel) and of the GRU memory update, pinned one after the
other (zt_set_kernel_choice) at the BASELINE configs' shapes, alone on the chip: microseconds per call (torch events
over 30 calls after 5 warm-up calls).  Where the library's switches between them come from.
    python tools/exp/p23_kernels.py [--cus N]   (N: run on a CU-masked stream of the chip's LAST N compute units)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import ctypes as C

import inputs as I
from helpers import build_tgn
from zebra_amd import _capi

if os.environ.get("ZT_LIB"):          # another build of the library (tools/build_variant.sh)
    _capi.LIB_PATH = os.path.join(ROOT, os.environ["ZT_LIB"])


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


def main():
    cus = int(sys.argv[sys.argv.index("--cus") + 1]) if "--cus" in sys.argv else 0
    stream = None
    if cus:
        h = C.c_void_p()
        tot = torch.cuda.get_device_properties(0).multi_processor_count
        _capi.check(_capi.lib().zt_stream_create_masked(C.byref(h), C.c_int32(tot - cus), C.c_int32(tot)))
        stream = torch.cuda.ExternalStream(h.value)
    D = T = 100
    N, E1 = 200000, 50000
    for F, k, rows_e, rows_g in ((1, 20, 12288, 8192), (1, 40, 3000, 2000), (172, 20, 1800, 1200), (172, 20, 600, 400)):
        g = torch.Generator().manual_seed(1)
        w = I.model_weights(D, F, T, 2, 9)
        _, efeat = I.random_tables(N, E1, D, F, 9)
        tgn = build_tgn(N, E1, D, F, T, k, [0.1, 0.1], [0.5, 0.95], w, efeat).eval()
        dev = tgn.device
        m = tgn.memory
        m.memory.copy_(torch.randn((N, D), generator=g).to(dev))
        m.messages.copy_(torch.randn((N, 2 * D + F + T), generator=g).to(dev))
        ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
        with ctx:
            # ---- output layers
            n = rows_e
            nodes = torch.randint(0, N, (n,), generator=g, dtype=torch.int32).to(dev)
            on = torch.randint(0, N, (2, n, k), generator=g, dtype=torch.int32).to(dev)
            oe = torch.randint(0, E1, (2, n, k), generator=g, dtype=torch.int32).to(dev)
            od = (torch.rand((2, n, k), generator=g) * 3.0e6).to(dev)
            ow = torch.rand((2, n, k), generator=g).to(dev)
            em = tgn.embedding_module
            res = {}
            for name, ch in (("tiled", _capi.OUT_TILED), ("latency", _capi.OUT_LATENCY), ("persist", _capi.OUT_PERSIST)):
                _capi.set_kernel_choice(_capi.CHOICE_EMBED_OUT, ch)
                _capi.lib().zt_profile_reset(); _capi.lib().zt_profile_enable(1)
                timed(lambda: em.embed_device(m.memory, nodes, on, oe, od, ow, memory_obj=m))
                _capi.lib().zt_profile_enable(0)
                c, ms = C.c_int64(), C.c_double()
                _capi.lib().zt_profile_read(b"embed_out", C.byref(c), C.byref(ms))
                res[name] = 1e3 * ms.value / max(1, c.value)
            _capi.set_kernel_choice(_capi.CHOICE_EMBED_OUT, 0)
            print("embed_out F=%d k=%d rows=%d: " % (F, k, n) + "  ".join("%s %.1f us" % kv for kv in res.items()), flush=True)
            # ---- GRU
            n = rows_g
            res = {}
            ids = (torch.randperm(N - 1, generator=g)[:n] + 1).to(torch.int32).to(dev)
            table = em._projection(m)
            for name, ch in (("tile", _capi.GRU_TILE), ("split", _capi.GRU_SPLIT)):
                _capi.set_kernel_choice(_capi.CHOICE_GRU, ch)

                def run():
                    m._flag_buf[ids.long()] = 1
                    tgn.memory_updater.update_device(m, ids, n)
                _capi.lib().zt_profile_reset(); _capi.lib().zt_profile_enable(1)
                timed(run)
                _capi.lib().zt_profile_enable(0)
                c, ms = C.c_int64(), C.c_double()
                _capi.lib().zt_profile_read(b"gru_update", C.byref(c), C.byref(ms))
                res[name] = 1e3 * ms.value / max(1, c.value)
            _capi.set_kernel_choice(_capi.CHOICE_GRU, 0)
            print("gru       F=%d rows=%d (incl. compaction of the flagged ids): " % (F, n) + "  ".join("%s %.1f us" % kv for kv in res.items()), flush=True)
        del tgn
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
