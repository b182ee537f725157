#!/bin/bash
# Two variants of the bench command, alternately on ONE box, twice each (the same command moves by up to 8 % from box to box,
# 1-3 % from run to run: every comparison quoted in DESIGN.md is made this way).  Run it through gpurun:
#   gpurun -- 'bash tools/exp/ab.sh NAME "<flags of A>" "<flags of B>" ["<common flags>"]'
# e.g.  bash tools/exp/ab.sh pairs "" "--chain-pairs" "--steps 200 --warmup 20 --legs c3 --cpu-edges 0 --no-score"
#       bash tools/exp/ab.sh c4cus "--tppr-cus 0" "--tppr-cus 64" "--workload c4 --steps 100 --warmup 10 --legs none"
# Results: gpurun_out/ab_NAME_{a,b}_{1,2}.json and a one-line summary of each (tools/exp/sb.py).
NAME=${1:?name}; A=$2; B=$3; COMMON=${4:---steps 200 --warmup 20 --legs none --cpu-edges 0 --no-score}
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for r in 1 2; do
  for v in a b; do
    F=$A; [ $v = b ] && F=$B
    timeout -k 10 300 python3 bench.py $COMMON $F > gpurun_out/ab_${NAME}_${v}_$r.json 2> gpurun_out/ab_${NAME}_${v}_$r.err || echo "variant $v run $r failed"
  done
done
for f in gpurun_out/ab_${NAME}_*.json; do python3 tools/exp/sb.py $f; done
