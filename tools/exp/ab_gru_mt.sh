# A/B on one box: k_gru with 32-row tiles from 4096 rows (default) or never (ZT_GRU_MT2_MIN_ROWS=1000000): C5, 200 steps
for rep in 1 2; do
  python bench.py --workload c5 --steps 200 --warmup 20 --legs none --cpu-edges 0 --no-score > gpurun_out/abg_mt2_$rep.json 2> gpurun_out/abg_mt2_$rep.err || exit 1
  ZT_GRU_MT2_MIN_ROWS=1000000 python bench.py --workload c5 --steps 200 --warmup 20 --legs none --cpu-edges 0 --no-score > gpurun_out/abg_mt1_$rep.json 2> gpurun_out/abg_mt1_$rep.err || exit 1
done
python tools/exp/sb.py gpurun_out/abg_*.json
