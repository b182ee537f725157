for t in 0 4; do
  ZT_STREAM_TUNE=$t python bench.py --workload c5 --steps 200 --cpu-edges 0 > gpurun_out/x_c5.json 2> gpurun_out/x.err || exit 1
  echo "tune=$t $(grep 'host enqueue' gpurun_out/x.err)"; python tools/showbench.py gpurun_out/x_c5.json | sed -n 2p | cut -c1-90
done
