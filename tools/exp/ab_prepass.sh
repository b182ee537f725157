# A/B on ONE box: C5, the driver's 20-step run and a 200-step run, with the prepass variants
#   new  = dependencies of big groups from the cooperative sort, single-workgroup form up to 4096 accesses
#   old  = ZT_DEPS_SORT_MIN=0 ZT_PREPASS_FUSED_MAX=12288 (round 3's prepass)
#   mix  = sort on, single-workgroup form up to 12288 accesses
for rep in 1 2; do
for v in new old mix; do
  case $v in
    new) E="";;
    old) E="ZT_DEPS_SORT_MIN=0 ZT_PREPASS_FUSED_MAX=12288";;
    mix) E="ZT_PREPASS_FUSED_MAX=12288";;
  esac
  env $E python bench.py --workload c5 --steps 20 --warmup 5 --legs none --cpu-edges 0 --no-score > gpurun_out/ab_${v}_20_$rep.json 2> gpurun_out/ab_${v}_20_$rep.err || exit 1
  env $E python bench.py --workload c5 --steps 200 --warmup 20 --legs none --cpu-edges 0 --no-score > gpurun_out/ab_${v}_200_$rep.json 2> gpurun_out/ab_${v}_200_$rep.err || exit 1
done
done
python tools/exp/sb.py gpurun_out/ab_*_20_*.json gpurun_out/ab_*_200_*.json
