import sys, os, ctypes as C
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from zebra_amd import _capi
_capi.LIB_PATH = '/root/repo/tools/out/libzebra_aggstamp.so'
sys.argv = ['bench.py', '--steps', '20', '--warmup', '3', '--prefill-steps', '60', '--no-pipeline', '--cpu-edges', '0', '--workload', sys.argv[1] if len(sys.argv) > 1 else 'c5']
import runpy
lib = _capi.lib()
try:
    runpy.run_path('/root/repo/bench.py', run_name='__main__')
except SystemExit:
    pass
a = np.zeros(8, np.uint64)
lib.zt_debug_agg(a.ctypes.data_as(C.c_void_p), C.c_int(0))
a = a.astype(np.float64)
names = ["scalars+normalise", "gather issue", "time encode", "LDS stores + pad", "barrier", "fc1 MFMA", "bias/relu -> LDS", "k-reduction + store"]
tot = a.sum()
wl = sys.argv[-1] if sys.argv[-1] in ("c2", "c3", "c4", "c5") else "c5"
for n, v in zip(names, a):
    print("%-22s %5.1f%%   %10.0f clk total" % (n, 100 * v / tot, v))
print("total %.3e clk (thread 0 of every workgroup, summed over workgroups and launches)" % tot)
