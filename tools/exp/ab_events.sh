for i in 1 2; do
  for f in 0 40000000 20000000; do
    ZT_EVENT_FLAGS=$f python bench.py --steps 200 --cpu-edges 0 > gpurun_out/ab_${f}_${i}.json 2> gpurun_out/ab.err || exit 1
    echo "flags=$f run=$i $(grep 'host enqueue' gpurun_out/ab.err)"
  done
done
