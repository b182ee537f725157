#!/bin/bash
# The rocprofv3 passes behind profiles/rN (run on the GPU box through gpurun; the summaries are copied to profiles/ by hand):
#   tools/profile_round.sh <workload> <steps> <warmup> <commit> [last_n]
# EVERY pass is the command that is timed -- CU masks on, the 10 % prefill, the same batches per T-PPR launch: the driver's
# `bench.py --steps 20 --warmup 5` for C5 -- first the kernel trace + stats, then SEPARATE --pmc passes (FETCH_SIZE /
# WRITE_SIZE / SQ counters; never --pmc together with a trace).  Counter collection is restricted to the kernels of the
# timed step (--kernel-include-regex): the prefill's prepass / cleanup / staging launches are not instrumented.  Each pass
# runs under its own timeout and keeps its stderr ($O/*.err): a pass that dies leaves its log, the others still run.
WL=${1:-c5}; STEPS=${2:-20}; WARM=${3:-5}; COMMIT=${4:-unknown}; LAST=${5:-10}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_r4_$WL
rm -rf $O && mkdir -p $O
CMD="bench.py --workload $WL --steps $STEPS --warmup $WARM --legs none --cpu-edges 0 --no-score"
RE='k_stream|k_fc1_agg|k_gru|k_embed_out|k_build_messages|k_pruned_topk|k_affinity'
timeout -k 10 900 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 $CMD > $O/bench_under_rocprof.json 2> $O/kt.err
echo "[profile] kernel trace rc=$?"
python3 profiles/summarize.py $O/kt/kt_kernel_trace.csv $STEPS > $O/kernel_trace_summary.txt 2>> $O/kt.err
cp $O/kt/kt_kernel_stats.csv $O/kernel_stats.csv 2>/dev/null
for C in FETCH_SIZE WRITE_SIZE; do
  T=$(echo $C | cut -c1 | tr A-Z a-z)
  timeout -k 10 900 rocprofv3 --pmc $C --kernel-include-regex "$RE" -d $O/$T -o $T --output-format csv -- python3 -X faulthandler $CMD --no-profile > $O/$T.log 2> $O/$T.err
  echo "[profile] $C rc=$?"
done
python3 profiles/make_pmc_summary.py $O/f/f_counter_collection.csv $O/w/w_counter_collection.csv $LAST $WL $O/f.log $COMMIT > $O/pmc_summary.json 2> $O/pmc_summary.err
timeout -k 10 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-include-regex "$RE" -d $O/sq -o sq --output-format csv -- python3 -X faulthandler $CMD --no-profile > $O/sq.log 2> $O/sq.err
echo "[profile] SQ rc=$?"
python3 profiles/make_sq_summary.py $O/sq/sq_counter_collection.csv $LAST $WL > $O/sq_summary.json 2> $O/sq_summary.err
ls -la $O/f $O/w $O/sq 2>/dev/null | head -20
rm -rf $O/f $O/w $O/sq $O/kt
for f in $O/*.err; do echo "== $f"; grep -v "simple_timer\|tool.cpp\|output_stream\|amdgpu.ids" $f | tail -15; done
head -16 $O/kernel_trace_summary.txt
