for i in 1 2; do
  python bench.py --steps 200 --cpu-edges 0 > gpurun_out/abp_p_$i.json 2> gpurun_out/ab.err || exit 1
  echo "profile run=$i $(grep 'host enqueue' gpurun_out/ab.err)"
  python bench.py --steps 200 --cpu-edges 0 --no-profile > gpurun_out/abp_n_$i.json 2> gpurun_out/ab.err || exit 1
  echo "noprofile run=$i $(grep 'host enqueue' gpurun_out/ab.err)"
done
ZT_STREAM_WGS_PER_CU=0.5 python bench.py --steps 200 --cpu-edges 0 > gpurun_out/abp_half.json 2> gpurun_out/ab.err || exit 1
echo "half grid $(grep 'host enqueue' gpurun_out/ab.err)"
python tools/showbench.py gpurun_out/abp_half.json
