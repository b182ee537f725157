# bench.py against another build of the library (tools/build_variant.sh, tools/build_*.sh):
#   python tools/exp/bench_lib.py tools/out/libzebra_NAME.so <bench.py arguments>
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from zebra_amd import _capi
_capi.LIB_PATH = os.path.join(ROOT, sys.argv[1]) if not os.path.isabs(sys.argv[1]) else sys.argv[1]
sys.argv = ["bench.py"] + sys.argv[2:]
import bench
bench.main()
