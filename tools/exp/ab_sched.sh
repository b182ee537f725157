# A/B on ONE box, C5: launch-group taper, chain budget, CU share (20-step and 200-step runs, two repetitions)
run() { # name steps warm env...
  n=$1; st=$2; w=$3; shift 3
  env "$@" python bench.py --workload c5 --steps $st --warmup $w --legs none --cpu-edges 0 --no-score $EXTRA > gpurun_out/abs_${n}_${st}_$rep.json 2> gpurun_out/abs_${n}_${st}_$rep.err || exit 1
}
for rep in 1 2; do
  EXTRA=""            run new 20 5 A=1
  EXTRA=""            run notaper 20 5 ZT_GROUP_TAPER=0
  EXTRA=""            run nobudget8 20 5 ZT_CHAIN_BUDGET=0 ZT_STREAM_CHAINS=8
  EXTRA="--tppr-cus 96" run r96 20 5 ZT_CHAIN_BUDGET=0 ZT_GROUP_TAPER=0
  EXTRA=""            run new 200 20 A=1
  EXTRA="--tppr-cus 96" run r96 200 20 ZT_CHAIN_BUDGET=0 ZT_GROUP_TAPER=0
done
python tools/exp/sb.py gpurun_out/abs_*_20_*.json gpurun_out/abs_*_200_*.json
