#!/bin/bash
# The rocprofv3 passes behind profiles/rN (run on the GPU box through gpurun; copies are made by hand afterwards).
#   kernel trace + stats of the default bench, then separate --pmc passes (FETCH_SIZE / WRITE_SIZE / SQ counters).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 bench.py --steps 200 > $O/bench_under_rocprof.json 2> $O/kt.err
python3 profiles/summarize.py $O/kt/kt_kernel_trace.csv 100 > $O/kernel_trace_summary.txt
cp $O/kt/kt_kernel_stats.csv $O/kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE -d $O/f -o f --output-format csv -- python3 bench.py --steps 40 --cpu-edges 0 --prefill-steps 600 > $O/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/w -o w --output-format csv -- python3 bench.py --steps 40 --cpu-edges 0 --prefill-steps 600 > $O/w.log 2>&1
python3 profiles/make_pmc_summary.py $O/f/f_counter_collection.csv $O/w/w_counter_collection.csv 40 c5 > $O/pmc_summary.json
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 -d $O/sq -o sq --output-format csv -- python3 bench.py --steps 40 --cpu-edges 0 --prefill-steps 600 > $O/sq.log 2>&1
python3 profiles/make_sq_summary.py $O/sq/sq_counter_collection.csv 40 c5 > $O/sq_summary.json
rm -rf $O/f $O/w $O/sq $O/kt/kt_kernel_trace.csv
head -12 $O/kernel_trace_summary.txt
