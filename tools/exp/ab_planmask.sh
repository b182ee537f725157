# the plan stream unmasked (default) or confined to the T-PPR stream's CUs (ZT_PLAN_MASKED=1): default run (C5 20 steps + legs), one box
for rep in 1 2; do
  python bench.py --steps 20 --warmup 5 --no-score > gpurun_out/abp_free_$rep.json 2> gpurun_out/abp_free_$rep.err || exit 1
  ZT_PLAN_MASKED=1 python bench.py --steps 20 --warmup 5 --no-score > gpurun_out/abp_masked_$rep.json 2> gpurun_out/abp_masked_$rep.err || exit 1
done
python tools/exp/sb.py gpurun_out/abp_*.json
