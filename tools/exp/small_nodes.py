# experiment: C5's batch shape over fewer nodes (is k_fc1_agg bound by the random gather from the 4.5 GB table?)
import sys, runpy
sys.path.insert(0, '/root/repo')
from zebra_amd import synth
n = int(sys.argv[1])
synth.WORKLOADS["c5"] = dict(synth.WORKLOADS["c5"], n_nodes=n)
sys.argv = ['bench.py', '--steps', '100', '--cpu-edges', '0', '--no-pipeline', '--workload', 'c5']
runpy.run_path('/root/repo/bench.py', run_name='__main__')
