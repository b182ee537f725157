#!/bin/bash
# The product library against a variant build (tools/build_variant.sh), alternately on ONE box, twice each:
#   gpurun -- 'bash tools/exp/ab_lib.sh NAME tools/out/libzebra_VARIANT.so "<bench flags>"'
# Results: gpurun_out/abl_NAME_{prod,var}_{1,2}.json and one summary line each.
NAME=${1:?name}; VAR=${2:?variant .so}; COMMON=${3:---steps 200 --warmup 20 --legs none --cpu-edges 0 --no-score --steady-steps 0}
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for r in 1 2; do
  timeout -k 10 300 python3 bench.py $COMMON > gpurun_out/abl_${NAME}_prod_$r.json 2> gpurun_out/abl_${NAME}_prod_$r.err || echo "prod run $r failed"
  timeout -k 10 300 python3 tools/exp/bench_lib.py $VAR $COMMON > gpurun_out/abl_${NAME}_var_$r.json 2> gpurun_out/abl_${NAME}_var_$r.err || echo "variant run $r failed"
done
for f in gpurun_out/abl_${NAME}_*.json; do python3 tools/exp/sb.py $f; done
