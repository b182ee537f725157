# run bench.py against an experimental library (tools/out/libzebra_x.so)
import sys, runpy
sys.path.insert(0, '/root/repo')
from zebra_amd import _capi
_capi.LIB_PATH = '/root/repo/tools/out/libzebra_x.so'
sys.argv = ['bench.py'] + sys.argv[1:]
runpy.run_path('/root/repo/bench.py', run_name='__main__')
