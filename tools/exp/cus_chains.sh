# C5, 200 steps: fewer compute units for the T-PPR stream with fewer hub chains (the main stream is the equal bound now)
for cfg in "96 16" "64 8" "64 12" "64 16" "96 8"; do
  set -- $cfg
  ZT_STREAM_CHAINS=$2 python bench.py --workload c5 --steps 200 --warmup 20 --legs none --cpu-edges 0 --no-score --tppr-cus $1 > gpurun_out/cc_$1_$2.json 2> gpurun_out/cc_$1_$2.err || exit 1
done
python tools/exp/sb.py gpurun_out/cc_*.json
