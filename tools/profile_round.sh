#!/bin/bash
# The rocprofv3 passes behind profiles/rN (run on the GPU box through gpurun; copies are made by hand afterwards):
#   tools/profile_round.sh <workload> <steps> <commit>
# kernel trace + stats of the bench, then SEPARATE --pmc passes (FETCH_SIZE / WRITE_SIZE / SQ counters), every pass the
# SAME command (same prefill, same batches per T-PPR launch) as the timed one; never --pmc together with a trace.
set -e
WL=${1:-c5}; STEPS=${2:-200}; COMMIT=${3:-unknown}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_$WL
rm -rf $O && mkdir -p $O
CMD="bench.py --workload $WL --steps $STEPS"
rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 $CMD > $O/bench_under_rocprof.json 2> $O/kt.err
python3 profiles/summarize.py $O/kt/kt_kernel_trace.csv $STEPS > $O/kernel_trace_summary.txt
cp $O/kt/kt_kernel_stats.csv $O/kernel_stats.csv
echo "[profile] kernel trace done"
# (counter passes run WITHOUT the CU masks: rocprofv3's counter mode crashes when k_fc1_agg_reg -- 512 registers per lane --
#  is launched on a CU-masked stream; bytes and instruction counts per launch do not depend on the mask, durations do and
#  are taken from the kernel trace above)
export ZT_BENCH_NO_MASKS=1
# (... and with a shorter prefill: with the bench's 10 % prefill -- 2441 batches, ~25 k launches before the timed region --
#  the counter mode segfaults inside a launch; the T-PPR rows are full long before batch 600 of the stream)
PMC="$CMD --cpu-edges 0 --prefill-steps ${PMC_PREFILL:-600} --steps 40"
rocprofv3 --pmc FETCH_SIZE -d $O/f -o f --output-format csv -- python3 $PMC > $O/f.log 2>&1
echo "[profile] FETCH_SIZE done"
rocprofv3 --pmc WRITE_SIZE -d $O/w -o w --output-format csv -- python3 $PMC > $O/w.log 2>&1
echo "[profile] WRITE_SIZE done"
python3 profiles/make_pmc_summary.py $O/f/f_counter_collection.csv $O/w/w_counter_collection.csv 40 $WL $O/f.log $COMMIT > $O/pmc_summary.json
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 -d $O/sq -o sq --output-format csv -- python3 $PMC > $O/sq.log 2>&1
python3 profiles/make_sq_summary.py $O/sq/sq_counter_collection.csv 40 $WL > $O/sq_summary.json
rm -rf $O/f $O/w $O/sq $O/kt/kt_kernel_trace.csv $O/kt
head -14 $O/kernel_trace_summary.txt
