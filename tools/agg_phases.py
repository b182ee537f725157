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
names = ["cons mfma", "cons epilogue", "prod work", "prod wait", "cons wait", "p:normalize", "p:issue", "p:cos"]
tot = a.sum()
import os
mt = int(os.environ.get("ZT_AGG_MT", "5"))
rq = (mt * 16) // 20
nwg = ((12288 + rq - 1) // rq) * 2 * 83   # = tiles          # prefill 60 + warmup 3 + 20 steps
for n, v in zip(names, a):
    print("%-22s %5.1f%%   %8.0f clk/WG" % (n, 100 * v / tot, v / nwg))
print("total %.0f clk/WG" % (tot / nwg))
