"""zt_affinity over batch sizes (the latency-organised kernel against the tiled one):
    python tools/exp/score_bench.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import ctypes as C
    import torch
    import inputs as I
    from helpers import build_tgn
    from zebra_amd import _capi
    D = T = 100
    w = I.model_weights(D, 1, T, 2, 5)
    _, efeat = I.random_tables(50, 60, D, 1, 5)
    tgn = build_tgn(50, 60, D, 1, T, 20, [0.1, 0.1], [0.5, 0.95], w, efeat).eval()
    lib = _capi.lib(); res = []
    for B in (200, 600, 1000, 2000, 4096, 8192):
        emb = torch.randn((3 * B, 300), device="cuda")
        for _ in range(5): tgn.score_device(emb)
        torch.cuda.synchronize(); lib.zt_profile_reset(); lib.zt_profile_enable(1)
        for _ in range(30): tgn.score_device(emb)
        torch.cuda.synchronize(); lib.zt_profile_enable(0)
        cnt, ms = C.c_int64(), C.c_double(); lib.zt_profile_read(b"score", C.byref(cnt), C.byref(ms))
        res.append("%d: %.1f" % (B, 1e3 * ms.value / max(1, cnt.value)))
    print("%-8s score us by B  %s" % (sys.argv[1], "  ".join(res)))
else:
    for name, env in (("latency", {"ZT_AFFINITY_TILED_MIN_B": "100000000"}), ("tiled", {"ZT_AFFINITY_TILED_MIN_B": "1"})):
        subprocess.run([sys.executable, __file__, name], env=dict(os.environ, **env), check=True)
