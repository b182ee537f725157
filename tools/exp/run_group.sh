timeout -k 10 500 python -m pytest tests/test_embed_gpu.py -x -q -m gpu -k "grouped or pipelined or full_size" > gpurun_out/t_small.txt 2>&1 || { tail -30 gpurun_out/t_small.txt; exit 1; }
tail -2 gpurun_out/t_small.txt
for i in 1 2; do
  timeout -k 10 200 python bench.py --workload c5 --steps 20 --warmup 5 --cpu-edges 0 2>&1 | grep "host enqueue"
done
timeout -k 10 200 python bench.py --workload c5 --steps 200 --cpu-edges 0 2>&1 | grep "host enqueue"
