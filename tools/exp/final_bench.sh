for c in c5 c2 c3 c4; do
  python bench.py --workload $c > gpurun_out/final_$c.json 2> gpurun_out/final_$c.err || exit 1
  echo "$c $(grep 'host enqueue' gpurun_out/final_$c.err)"
done
python bench.py --steps 20 --warmup 5 > gpurun_out/final_c5_20steps.json 2> gpurun_out/final_c5_20.err || exit 1
echo "c5 20 steps $(grep 'host enqueue' gpurun_out/final_c5_20.err)"
