for i in 1 2; do for g in 1 2; do
  timeout -k 10 200 python bench.py --workload c5 --steps 20 --warmup 5 --cpu-edges 0 --group $g > gpurun_out/x_c5.json 2> gpurun_out/x.err || { tail -5 gpurun_out/x.err; exit 1; }
  echo "20 steps group=$g $(grep 'host enqueue' gpurun_out/x.err)"
done; done
for g in 1 2; do
  timeout -k 10 200 python bench.py --workload c3 --steps 200 --cpu-edges 0 --group $g > gpurun_out/x_c3.json 2> gpurun_out/x.err || { tail -5 gpurun_out/x.err; exit 1; }
  echo "c3 group=$g $(grep 'host enqueue' gpurun_out/x.err)"
  timeout -k 10 200 python bench.py --workload c2 --steps 200 --cpu-edges 0 --group $g > gpurun_out/x_c2.json 2> gpurun_out/x.err || { tail -5 gpurun_out/x.err; exit 1; }
  echo "c2 group=$g $(grep 'host enqueue' gpurun_out/x.err)"
done
timeout -k 10 200 python bench.py --workload c3 --steps 200 --cpu-edges 0 --group 4 2>&1 | grep "host enqueue"
timeout -k 10 200 python bench.py --workload c2 --steps 200 --cpu-edges 0 --group 4 2>&1 | grep "host enqueue"
