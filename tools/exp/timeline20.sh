# Timeline of the driver's timed region (bench.py --steps 20 --warmup 5, C5 only) from a rocprofv3 kernel trace:
# every kernel of every queue from the end of the warm-up to the end of the region.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/tl20; rm -rf $O; mkdir -p $O
STEPS=${1:-20}
timeout -k 10 600 rocprofv3 --kernel-trace -d $O/kt -o kt --output-format csv -- python3 bench.py --workload c5 --steps $STEPS --warmup 5 --legs none --cpu-edges 0 --no-score --no-profile --steady-steps 0 $2 > $O/bench.json 2> $O/kt.err
python3 tools/exp/timeline20.py $O/kt/kt_kernel_trace.csv $STEPS > $O/timeline.txt
rm -rf $O/kt
tail -5 $O/timeline.txt
