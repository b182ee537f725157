# C5, 200 steps: T-PPR CU share in steps of a quarter XCD (the persistent aggregation kernel strides its tiles over its
# workgroups: does a mask that splits an XCD still leave a straggler?)
for c in 96 88 80 72 96 80; do
  python bench.py --workload c5 --steps 200 --warmup 20 --legs none --cpu-edges 0 --no-score --tppr-cus $c > gpurun_out/cs_$c.json 2> gpurun_out/cs_$c.err || exit 1
  python tools/exp/sb.py gpurun_out/cs_$c.json
done
