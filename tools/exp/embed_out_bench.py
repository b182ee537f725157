"""k_embed_out (throughput-organised) against k_embed_out2 (latency-organised) over batch sizes, F = 1 and F = 172:
    python tools/exp/embed_out_bench.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import ctypes as C
    import numpy as np, torch
    import inputs as I
    from helpers import build_tgn
    from zebra_amd import _capi
    F = int(sys.argv[2]); D = T = 100; k = 20; N = 200000; E1 = 100000
    g = torch.Generator().manual_seed(5)
    w = I.model_weights(D, F, T, 2, 77)
    efeat = torch.randn((E1, F), generator=g)
    tgn = build_tgn(N, E1, D, F, T, k, [0.1, 0.1], [0.5, 0.95], w, efeat).eval()
    dev = tgn.device
    tgn.memory.memory.copy_(torch.randn((N, D), generator=g).to(dev))
    lib = _capi.lib(); em = tgn.embedding_module
    res = []
    for n in (600, 1200, 1800, 3000, 6000, 12288):
        nodes = torch.randint(0, N, (n,), generator=g, dtype=torch.int32).to(dev)
        args = [t.to(dev).contiguous() for t in (torch.randint(0, N, (2, n, k), generator=g, dtype=torch.int32),
                torch.randint(0, E1, (2, n, k), generator=g, dtype=torch.int32), torch.rand((2, n, k), generator=g) * 3e6, torch.rand((2, n, k), generator=g))]
        for _ in range(5): em.embed_device(tgn.memory.memory, nodes, *args, memory_obj=tgn.memory)
        torch.cuda.synchronize(); lib.zt_profile_reset(); lib.zt_profile_enable(1)
        for _ in range(30): em.embed_device(tgn.memory.memory, nodes, *args, memory_obj=tgn.memory)
        torch.cuda.synchronize(); lib.zt_profile_enable(0)
        cnt, ms = C.c_int64(), C.c_double(); lib.zt_profile_read(b"embed_out", C.byref(cnt), C.byref(ms))
        res.append("%d: %.1f" % (n, 1e3 * ms.value / max(1, cnt.value)))
    print("%-10s F=%-3d embed_out us by rows  %s" % (sys.argv[1], F, "  ".join(res)))
else:
    for F in ("1", "172"):
        for name, env in (("out2", {"ZT_EMBED_OUT2_MAX_ROWS": "1000000"}), ("out", {"ZT_EMBED_OUT2_MAX_ROWS": "0"})):
            subprocess.run([sys.executable, __file__, name, F], env=dict(os.environ, **env), check=True)
