# C5, the driver's 20-step run: batches per T-PPR launch (the taper queries the last three batches one by one in any case)
for rep in 1 2; do
for g in 2 3 4; do
  python bench.py --workload c5 --steps 20 --warmup 5 --legs none --cpu-edges 0 --no-score --group $g > gpurun_out/g20_${g}_$rep.json 2> gpurun_out/g20_${g}_$rep.err || exit 1
done
done
python tools/exp/sb.py gpurun_out/g20_*.json
