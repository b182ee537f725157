for k in 10 20 40 80 160; do
  python bench.py --steps $k --warmup 5 --cpu-edges 0 2> gpurun_out/x.err > gpurun_out/x.json || exit 1
  echo "steps=$k $(grep 'host enqueue' gpurun_out/x.err)"
done
