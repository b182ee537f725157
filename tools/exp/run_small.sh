python -m pytest tests/test_embed_gpu.py tests/test_distributed_gpu.py -x -q -m gpu > gpurun_out/t_small.txt 2>&1 || { tail -20 gpurun_out/t_small.txt; exit 1; }
tail -2 gpurun_out/t_small.txt
for c in c2 c3 c4 c5; do
  python bench.py --workload $c --steps 200 --cpu-edges 0 > gpurun_out/x_$c.json 2> gpurun_out/x.err || exit 1
  echo "$c $(grep 'host enqueue' gpurun_out/x.err)"; python tools/showbench.py gpurun_out/x_$c.json | sed -n 2p
done
python bench.py --workload c5 --steps 20 --warmup 5 --cpu-edges 0 > gpurun_out/x_c5_20.json 2> gpurun_out/x.err || exit 1
echo "c5 20 steps $(grep 'host enqueue' gpurun_out/x.err)"
