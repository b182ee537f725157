# five hardware queues (default) or four (ZT_FOUR_QUEUES=1: plan stream = message stream, messages enqueued first): default run, one box
for rep in 1 2; do
  python bench.py --steps 20 --warmup 5 --no-score > gpurun_out/ab4_five_$rep.json 2> gpurun_out/ab4_five_$rep.err || exit 1
  ZT_FOUR_QUEUES=1 python bench.py --steps 20 --warmup 5 --no-score > gpurun_out/ab4_four_$rep.json 2> gpurun_out/ab4_four_$rep.err || exit 1
done
python bench.py --workload c5 --steps 200 --legs none --cpu-edges 0 --no-score > gpurun_out/ab4_five_200.json 2> gpurun_out/ab4_five_200.err
ZT_FOUR_QUEUES=1 python bench.py --workload c5 --steps 200 --legs none --cpu-edges 0 --no-score > gpurun_out/ab4_four_200.json 2> gpurun_out/ab4_four_200.err
python tools/exp/sb.py gpurun_out/ab4_*.json
