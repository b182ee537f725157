"""Times k_fc1_agg_wide (F = 172) and k_embed_out alone, whole chip, at C2's (600 rows) and C3's (1800 rows) batch shape:
    python tools/exp/wide_bench.py [rows ...]
With a -DZT_WIDE_STAMP build of aggregate_wide.hip (ZT_EXTRA_HIPFLAGS=-DZT_WIDE_STAMP python -m zebra_amd.build after touching
the file) it also prints where a wave's cycles go: top of the tiles / MFMA groups / epilogues."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import ctypes as C
import numpy as np, torch
import inputs as I
from helpers import build_tgn
from zebra_amd import _capi
D = T = 100; F = 172; k = int(os.environ.get("WIDE_K", "20")); N = 10985; E1 = 672448
g = torch.Generator().manual_seed(5)
w = I.model_weights(D, F, T, 2, 77)
efeat = torch.randn((E1, F), generator=g)
tgn = build_tgn(N, E1, D, F, T, k, [0.1, 0.1], [0.5, 0.95], w, efeat).eval()
dev = tgn.device
tgn.memory.memory.copy_(torch.randn((N, D), generator=g).to(dev))
lib = _capi.lib()
em = tgn.embedding_module
for n in [int(x) for x in sys.argv[1:]] or [600, 1800]:
    nodes = torch.randint(0, N, (n,), generator=g, dtype=torch.int32).to(dev)
    on = torch.randint(0, N, (2, n, k), generator=g, dtype=torch.int32)
    oe = torch.randint(0, E1, (2, n, k), generator=g, dtype=torch.int32)
    od = torch.rand((2, n, k), generator=g) * 3.0e6
    ow = torch.rand((2, n, k), generator=g)
    args = [t.to(dev).contiguous() for t in (on, oe, od.float(), ow.float())]
    for _ in range(5):
        em.embed_device(tgn.memory.memory, nodes, *args, memory_obj=tgn.memory)
    torch.cuda.synchronize()
    lib.zt_profile_reset(); lib.zt_profile_enable(1)
    for _ in range(30):
        em.embed_device(tgn.memory.memory, nodes, *args, memory_obj=tgn.memory)
    torch.cuda.synchronize()
    lib.zt_profile_enable(0)
    out = []
    for name in (b"fc1_agg", b"embed_out"):
        cnt, ms = C.c_int64(), C.c_double()
        lib.zt_profile_read(name, C.byref(cnt), C.byref(ms))
        out.append("%s %.1f us" % (name.decode(), 1e3 * ms.value / max(1, cnt.value)))
    tiles = (n * k + 15) // 16 * 2
    fl = 2.0 * n * 2 * k * (F + T) * D
    us = float(out[0].split()[1])
    print("rows %d (%d M-tiles): %s   %.1f TF/s = %.0f %% of the f32 MFMA peak" % (n, tiles, ", ".join(out), fl / us / 1e6, 100 * fl / us / 1e6 / 157.3))
    if hasattr(lib, "zt_debug_wide"):
        buf = (C.c_ulonglong * (16 * 1024))()
        lib.zt_debug_wide(buf)
        full = np.array(list(buf), dtype=np.float64)
        a = full[:8192].reshape(1024, 8)
        p2 = full[8192:].reshape(1024, 8)[a[:, 3] > 0]
        a = a[a[:, 3] > 0]
        t0 = a[:, 4].min()
        print("   waves %d: tiles/wave %.2f (max %d); per tile: top %.0f, mfma %.0f, epilogue %.0f cycles; prologue %.0f cycles; wave %.1f us (max %.1f), "
              "start spread %.1f us, %.2f GHz" % (len(a), a[:, 3].mean(), a[:, 3].max(), (a[:, 0] / a[:, 3]).mean(), (a[:, 1] / a[:, 3]).mean(),
                                               (a[:, 2] / a[:, 3]).mean(), a[:, 6].mean(), ((a[:, 5] - a[:, 4]) * 0.01).mean(),
                                               ((a[:, 5] - t0) * 0.01).max(), ((a[:, 4] - t0) * 0.01).max(),
                                               np.median(a[:, 7] / ((a[:, 5] - a[:, 4]) * 1e-8)) / 1e9))
        print("   prologue stamps (cycles from wave start; mean): fetch issued %.0f, all loads issued %.0f, first scalars settled %.0f, LDS written %.0f, "
              "barrier passed %.0f, prep done %.0f" % tuple(p2[:, q].mean() for q in range(6)))
