# how many hardware queues: the plan stream of its own (default) or on the message stream (ZT_PLAN_ON_MSG=1); GPU_MAX_HW_QUEUES
for rep in 1 2; do
  python bench.py --steps 20 --warmup 5 --cpu-edges 0 --no-score > gpurun_out/abq_own_$rep.json 2> gpurun_out/abq_own_$rep.err || exit 1
  ZT_PLAN_ON_MSG=1 python bench.py --steps 20 --warmup 5 --cpu-edges 0 --no-score > gpurun_out/abq_shared_$rep.json 2> gpurun_out/abq_shared_$rep.err || exit 1
  GPU_MAX_HW_QUEUES=2 python bench.py --steps 20 --warmup 5 --cpu-edges 0 --no-score > gpurun_out/abq_hwq2_$rep.json 2> gpurun_out/abq_hwq2_$rep.err || exit 1
done
python tools/exp/sb.py gpurun_out/abq_*.json
