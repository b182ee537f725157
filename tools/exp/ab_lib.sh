# A/B on one box: the library in the tree against tools/out/libzebra_x.so (tools/exp/bench_x.py), alternately, twice:
#   bash tools/exp/ab_lib.sh "<bench args>" <tag>
ARGS=${1:---workload c5 --steps 200 --warmup 20 --legs none --cpu-edges 0 --no-score}; TAG=${2:-abl}
for rep in 1 2; do
  python bench.py $ARGS > gpurun_out/${TAG}_new_$rep.json 2> gpurun_out/${TAG}_new_$rep.err || exit 1
  python tools/exp/bench_x.py $ARGS > gpurun_out/${TAG}_old_$rep.json 2> gpurun_out/${TAG}_old_$rep.err || exit 1
done
python tools/exp/sb.py gpurun_out/${TAG}_*.json
