# more hub chains / a lower threshold, per workload and T-PPR CU share (round 4; builds variants of the library in place,
# ends with the product build):   bash tools/exp/chains_exp.sh
cd "$GRAFT_REPO_ROOT"
for cfg in "16 24" "16 12" "32 12"; do
  set -- $cfg
  touch zebra_amd/csrc/tppr_state.hpp
  ZT_EXTRA_HIPFLAGS="-DZT_MAX_CHAINS=$1 -DZT_HOT_MIN=$2" python -m zebra_amd.build > /dev/null 2>&1
  for w in "c3 32" "c3 64" "c3 96" "c2 32" "c2 64" "c5 96"; do
    set -- $cfg $w
    python bench.py --workload $3 --steps 200 --warmup 20 --cpu-edges 0 --no-score --legs none --tppr-cus $4 > gpurun_out/ch_$1_$2_$3_$4.json 2>/dev/null
    python - <<PY
import json
d=json.load(open("gpurun_out/ch_$1_$2_$3_$4.json"))
print("chains $1 hot_min $2 $3 cus $4: %.3f ms/step  k_stream %.0f us  fc1 %.0f" % (d["ms_per_step"], d["kernels"]["tppr_stream"]["avg_us"], d["kernels"]["fc1_agg"]["avg_us"]))
PY
  done
done
touch zebra_amd/csrc/tppr_state.hpp; python -m zebra_amd.build > /dev/null 2>&1
