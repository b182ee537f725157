for c in 56 64 72 80; do
  python bench.py --steps 200 --cpu-edges 0 --tppr-cus $c > gpurun_out/cus_$c.json 2> gpurun_out/cus.err || exit 1
  echo "tppr_cus=$c $(grep 'host enqueue' gpurun_out/cus.err)"; python tools/showbench.py gpurun_out/cus_$c.json | sed -n 2p
done
for c in 64 80; do
  python bench.py --steps 20 --warmup 5 --cpu-edges 0 --tppr-cus $c > gpurun_out/cus20_$c.json 2> gpurun_out/cus.err || exit 1
  echo "20 steps tppr_cus=$c $(grep 'host enqueue' gpurun_out/cus.err)"
done
for c in 0 32 64; do
  python bench.py --workload c3 --steps 200 --cpu-edges 0 --tppr-cus $c > gpurun_out/cus3_$c.json 2> gpurun_out/cus.err || exit 1
  echo "c3 tppr_cus=$c $(grep 'host enqueue' gpurun_out/cus.err)"; python tools/showbench.py gpurun_out/cus3_$c.json | sed -n 2p
done
