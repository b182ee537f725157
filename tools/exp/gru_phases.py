"""Phases of k_gru at a small batch (a -DZT_GRU_STAMP build of memory_update.hip): shader clocks of wave 0 between the
phase boundaries, averaged over the workgroups of the last launch.   python tools/exp/gru_phases.py [rows] [F]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import ctypes as C
import numpy as np, torch
import inputs as I
from helpers import build_tgn
from zebra_amd import _capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
F = int(sys.argv[2]) if len(sys.argv) > 2 else 172
D = T = 100; N = 20000; E1 = 1000
w = I.model_weights(D, F, T, 2, 7)
_, efeat = I.random_tables(N, E1, D, F, 7)
tgn = build_tgn(N, E1, D, F, T, 20, [0.1, 0.1], [0.5, 0.95], w, efeat).eval()
lib = _capi.lib()
m = tgn.memory
g = torch.Generator().manual_seed(1)
m.messages.copy_(torch.randn(m.messages.shape, generator=g).cuda()); m.memory.copy_(torch.randn(m.memory.shape, generator=g).cuda() * 0.1)
ids = torch.randperm(N - 1, generator=g)[:n].to(torch.int32).cuda() + 1
lib.zt_profile_reset(); lib.zt_profile_enable(1)
for it in range(20):
    m._flag_buf.zero_(); m._flag_buf[ids.long()] = 1
    tgn.memory_updater.update_device(m, ids, n)
torch.cuda.synchronize(); lib.zt_profile_enable(0)
cnt, ms = C.c_int64(), C.c_double(); lib.zt_profile_read(b"gru_update", C.byref(cnt), C.byref(ms))
print("rows %d F %d: gru_update (select + k_gru) %.1f us" % (n, F, 1e3 * ms.value / cnt.value))
if hasattr(lib, "zt_debug_gru"):
    buf = (C.c_ulonglong * 20480)(); lib.zt_debug_gru(buf)
    a = np.array(list(buf), dtype=np.float64).reshape(2048, 10)[: (n + 15) // 16]
    names = ["n_rows read", "weights + ids issued, barrier", "tile staged", "gate products", "partials to LDS + barrier", "gates, arrival", "(last workgroup) commit"]
    a = a[a[:, 6] > 0]
    d = np.diff(a[:, :8], axis=1)
    print("  cycles per phase (mean over %d workgroups): " % len(a) + "; ".join("%s %.0f" % (nm, x) for nm, x in zip(names, d.mean(axis=0))) + "; to the arrival %.0f" % (a[:, 6] - a[:, 0]).mean())
    print("  workgroup start spread %.0f cycles, arrival spread %.0f" % (a[:, 0].max() - a[:, 0].min(), a[:, 6].max() - a[:, 6].min()))
