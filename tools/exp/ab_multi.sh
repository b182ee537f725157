#!/bin/bash
# the product library and several variant builds, round robin on ONE box, twice each:
#   gpurun -- 'bash tools/exp/ab_multi.sh NAME "<bench flags>" tools/out/libzebra_A.so tools/out/libzebra_B.so ...'
NAME=${1:?name}; COMMON=$2; shift 2
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for r in 1 2; do
  timeout -k 10 300 python3 bench.py $COMMON > gpurun_out/abm_${NAME}_prod_$r.json 2> gpurun_out/abm_${NAME}_prod_$r.err || echo "prod run $r failed"
  for V in "$@"; do
    b=$(basename $V .so); b=${b#libzebra_}
    timeout -k 10 300 python3 tools/exp/bench_lib.py $V $COMMON > gpurun_out/abm_${NAME}_${b}_$r.json 2> gpurun_out/abm_${NAME}_${b}_$r.err || echo "$b run $r failed"
  done
done
for f in gpurun_out/abm_${NAME}_*.json; do python3 tools/exp/sb.py $f; done
