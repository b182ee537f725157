# A/B on one box: the wait for the message build between aggregation and output layer (default) or in front of the GRU
for rep in 1 2; do
for wl in c2 c3 c4 c5; do
  python bench.py --workload $wl --cpu-edges 0 --no-score --legs none > gpurun_out/abw_${wl}_early_$rep.json 2> gpurun_out/abw_${wl}_early_$rep.err || exit 1
  ZT_EARLY_MSG_WAIT=0 python bench.py --workload $wl --cpu-edges 0 --no-score --legs none > gpurun_out/abw_${wl}_late_$rep.json 2> gpurun_out/abw_${wl}_late_$rep.err || exit 1
done
done
python tools/exp/sb.py gpurun_out/abw_*.json
