# message kernel: two positions per wave (default) or one (ZT_MSG_TWO_PER_WAVE=0); default run, one box
for rep in 1 2; do
  python bench.py --steps 20 --warmup 5 --no-score > gpurun_out/abm_two_$rep.json 2> gpurun_out/abm_two_$rep.err || exit 1
  ZT_MSG_TWO_PER_WAVE=0 python bench.py --steps 20 --warmup 5 --no-score > gpurun_out/abm_one_$rep.json 2> gpurun_out/abm_one_$rep.err || exit 1
done
python bench.py --workload c5 --steps 200 --legs none --cpu-edges 0 --no-score > gpurun_out/abm_two_200.json 2> gpurun_out/abm_two_200.err
ZT_MSG_TWO_PER_WAVE=0 python bench.py --workload c5 --steps 200 --legs none --cpu-edges 0 --no-score > gpurun_out/abm_one_200.json 2> gpurun_out/abm_one_200.err
python tools/exp/sb.py gpurun_out/abm_*.json
