# C5, the driver's 20-step run: T-PPR CU share x hub chains
for cfg in "96 16" "64 8" "64 6" "64 5" "64 10"; do
  set -- $cfg
  for rep in 1 2; do
  ZT_STREAM_CHAINS=$2 python bench.py --workload c5 --steps 20 --warmup 5 --legs none --cpu-edges 0 --no-score --tppr-cus $1 > gpurun_out/cc20_$1_$2_$rep.json 2> gpurun_out/cc20_$1_$2_$rep.err || exit 1
  done
done
for cfg in "64 6" "64 5"; do
  set -- $cfg
  ZT_STREAM_CHAINS=$2 python bench.py --workload c5 --steps 200 --warmup 20 --legs none --cpu-edges 0 --no-score --tppr-cus $1 > gpurun_out/cc_$1_$2.json 2> gpurun_out/cc_$1_$2.err || exit 1
done
python tools/exp/sb.py gpurun_out/cc20_*.json gpurun_out/cc_64_6.json gpurun_out/cc_64_5.json
