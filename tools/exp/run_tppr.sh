python -m pytest tests/test_tppr_gpu.py tests/test_embed_gpu.py -x -q -m gpu > gpurun_out/t_small.txt 2>&1 || { tail -20 gpurun_out/t_small.txt; exit 1; }
tail -2 gpurun_out/t_small.txt
for i in 1 2; do
python bench.py --workload c5 --steps 200 --cpu-edges 0 > gpurun_out/x_c5.json 2> gpurun_out/x.err || exit 1
echo "c5 $(grep 'host enqueue' gpurun_out/x.err)"; python tools/showbench.py gpurun_out/x_c5.json | sed -n 2p | cut -c1-100
done
