"""Times the aggregation kernel alone at C5's shape (12288 rows x 2 models x k = 20, 100 K nodes) for the diagnostic
variants of k_fc1_agg_reg (ZT_AGG_DBG: 1 no cosine, 2 no epilogue, 4 no projected rows) and for the older kernels.
    python tools/exp/agg_variants.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import numpy as np, torch
    import inputs as I
    from helpers import build_tgn
    D = T = 100; F = 1; k = 20; N = 1_000_000; E1 = 1000; n = 12288
    g = torch.Generator().manual_seed(5)
    w = I.model_weights(D, F, T, 2, 77)
    efeat = np.zeros((E1, 1), np.float32)
    tgn = build_tgn(N, E1, D, F, T, k, [0.1, 0.1], [0.5, 0.95], w, efeat).eval()
    dev = tgn.device
    tgn.memory.memory.copy_(torch.randn((N, D), generator=g).to(dev))
    nodes = torch.randint(0, N, (n,), generator=g, dtype=torch.int32).to(dev)
    on = torch.randint(0, N, (2, n, k), generator=g, dtype=torch.int32)
    oe = torch.randint(0, E1, (2, n, k), generator=g, dtype=torch.int32)
    od = torch.rand((2, n, k), generator=g) * 3.0e6
    ow = torch.rand((2, n, k), generator=g)
    args = [t.to(dev).contiguous() for t in (on, oe, od.float(), ow.float())]
    em = tgn.embedding_module
    from zebra_amd import _capi
    import ctypes as C
    lib = _capi.lib()
    for _ in range(5):
        em.embed_device(tgn.memory.memory, nodes, *args, memory_obj=tgn.memory)
    torch.cuda.synchronize()
    lib.zt_profile_reset(); lib.zt_profile_enable(1)
    for _ in range(30):
        em.embed_device(tgn.memory.memory, nodes, *args, memory_obj=tgn.memory)
    torch.cuda.synchronize()
    cnt, ms = C.c_int64(), C.c_double()
    lib.zt_profile_read(b"fc1_agg", C.byref(cnt), C.byref(ms))
    clk = (C.c_ulonglong * 4096)()
    lib.zt_debug_regclk(clk)
    a = np.array(list(clk), dtype=np.float64).reshape(1024, 4)
    msg = ""
    if a[:, 2].max() > 0:
        t0 = a[:, 1].min()
        st, en, cyc = (a[:, 1] - t0) * 0.01, (a[:, 2] - t0) * 0.01, a[:, 0]
        fill = (a[:, 1] - a[:, 3]) * 0.01
        msg = ("  waves: start %.1f..%.1f us, end %.1f..%.1f (median %.1f) us, run median %.1f max %.1f us, %.2f GHz; weight fill (entry -> first tile) median %.1f max %.1f us, first entry %.1f us before the first start"
               % (st.min(), st.max(), en.min(), en.max(), np.median(en), np.median(en - st), (en - st).max(),
                  np.median(cyc / ((en - st) * 1e-6)) / 1e9, np.median(fill), fill.max(), (t0 - a[:, 3].min()) * 0.01))
    print("%s: fc1_agg %.1f us%s" % (sys.argv[1], 1e3 * ms.value / max(1, cnt.value), msg))
else:
    for name, env in (("reg", {}), ("reg timed (all parts)", {"ZT_AGG_DBG": "8"}), ("reg mfma + bare epilogue only", {"ZT_AGG_DBG": "7"})):
        subprocess.run([sys.executable, __file__, name], env=dict(os.environ, **env), check=True)
