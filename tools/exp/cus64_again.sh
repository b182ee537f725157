# C5 on two XCDs for the T-PPR stream again, now with five hardware queues (the chain budget gives 5 chains there): 200 and 20 steps, one box
for rep in 1 2; do
  for c in 96 64; do
    python bench.py --workload c5 --steps 200 --warmup 20 --legs none --cpu-edges 0 --no-score --tppr-cus $c > gpurun_out/c64_${c}_200_$rep.json 2> gpurun_out/c64_${c}_200_$rep.err || exit 1
    python bench.py --workload c5 --steps 20 --warmup 5 --legs none --no-score --tppr-cus $c > gpurun_out/c64_${c}_20_$rep.json 2> gpurun_out/c64_${c}_20_$rep.err || exit 1
  done
done
ZT_STREAM_CHAINS=8 ZT_CHAIN_BUDGET=0 python bench.py --workload c5 --steps 200 --warmup 20 --legs none --cpu-edges 0 --no-score --tppr-cus 64 > gpurun_out/c64_64c8_200_1.json 2> gpurun_out/c64_64c8_200_1.err
python tools/exp/sb.py gpurun_out/c64_*.json
