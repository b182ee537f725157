# k_stream compiled for three waves per SIMD (168 VGPRs, some spills) but still launched with eight waves: what do the spills cost?
cd "$GRAFT_REPO_ROOT"
for b in 512 768; do
  touch zebra_amd/csrc/tppr_stream.hip
  ZT_EXTRA_HIPFLAGS="-DZT_STREAM_BOUNDS=$b" python -m zebra_amd.build > /dev/null 2>&1
  for w in c5 c3; do
    python bench.py --workload $w --steps 200 --warmup 20 --cpu-edges 0 --no-score --legs none > gpurun_out/bd_${b}_$w.json 2>/dev/null
    python - <<PY
import json
d=json.load(open("gpurun_out/bd_${b}_$w.json"))
print("bounds $b $w: %.4f ms/step  k_stream %.0f us" % (d["ms_per_step"], d["kernels"]["tppr_stream"]["avg_us"]))
PY
  done
done
touch zebra_amd/csrc/tppr_stream.hip; python -m zebra_amd.build > /dev/null 2>&1
