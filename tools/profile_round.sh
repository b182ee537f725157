#!/bin/bash
# The rocprofv3 passes behind profiles/rN (run on the GPU box through gpurun; the summaries are copied to profiles/ by hand):
#   tools/profile_round.sh <workload> <steps> <warmup> <commit> [last_n] [crash]
# EVERY pass is the command that is timed -- CU masks on, the 10 % prefill, the same batches per T-PPR launch: the driver's
# `bench.py --steps 20 --warmup 5` for C5 -- first the kernel trace + stats, then SEPARATE --pmc passes (FETCH_SIZE /
# WRITE_SIZE / SQ counters; never --pmc together with a trace).  Counter collection is restricted by kernel name
# (--kernel-include-regex): FETCH_SIZE first with the prepass kernels of the timed region included (RE_WIDE), and, if that
# pass dies, again with the timed step's kernels only (RE); WRITE_SIZE and the SQ pass with RE.  `crash` as sixth argument:
# one more FETCH_SIZE pass with EVERY dispatch instrumented and the process's memory map dumped (tools/symbolise.py), to
# put names to the frames of the rocprofv3 counter-mode crash.  Each pass runs under its own timeout and keeps its stderr.
WL=${1:-c5}; STEPS=${2:-20}; WARM=${3:-5}; COMMIT=${4:-unknown}; LAST=${5:-10}; CRASH=${6:-}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_r6_$WL
rm -rf $O && mkdir -p $O
CMD="bench.py --workload $WL --steps $STEPS --warmup $WARM --legs none --cpu-edges 0 --no-score --steady-steps 0"
# Counter passes SERIALISE kernels: a kernel that waits for a kernel of another stream -- the gate that releases a T-PPR launch
# group to the aggregation batch by batch (pipeline.hip) -- never sees it run (the first attempt, with hipStreamWaitValue32 as
# the gate, hung; the bounded gate gives up after 4 s per step).  The counter passes therefore release by launch, in the
# launch groups the timed run uses (C5: 4 batches, C2 / C3: 8); the kernel trace above them is the timed configuration itself.
# (ZT_RELEASE_LAUNCH_FULL: no taper; the first batch of a region goes alone, so 1 + n full groups: 21 steps for C5, 97 for the others)
case $WL in c5) PSTEPS=21; PLAST=5;; c2|c3|c1) PSTEPS=97; PLAST=10;; *) PSTEPS=$STEPS; PLAST=$LAST;; esac
PMC_CMD="bench.py --workload $WL --steps $PSTEPS --warmup $WARM --legs none --cpu-edges 0 --no-score --steady-steps 0 --release-by-launch-full --no-profile"
RE='k_stream|k_fc1_agg|k_gru|k_out_gru|k_embed_out|k_build_messages|k_last_pos|k_pruned_topk|k_affinity'
RE_WIDE="$RE|k_deps|k_own|k_reserve|k_hot_select|k_count|k_fill|k_hubacc|k_prepass_fused|k_cleanup"
timeout -k 10 900 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 $CMD > $O/bench_under_rocprof.json 2> $O/kt.err
echo "[profile] kernel trace rc=$?"
python3 profiles/summarize.py $O/kt/kt_kernel_trace.csv $STEPS > $O/kernel_trace_summary.txt 2>> $O/kt.err
cp $O/kt/kt_kernel_stats.csv $O/kernel_stats.csv 2>/dev/null
timeout -k 10 900 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "$RE_WIDE" -d $O/f -o f --output-format csv -- python3 -X faulthandler $PMC_CMD > $O/f.log 2> $O/f_wide.err
RC=$?; echo "[profile] FETCH_SIZE (timed step + prepass kernels) rc=$RC"
if [ $RC -ne 0 ]; then
  rm -rf $O/f
  timeout -k 10 900 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "$RE" -d $O/f -o f --output-format csv -- python3 -X faulthandler $PMC_CMD > $O/f.log 2> $O/f.err
  echo "[profile] FETCH_SIZE (timed step only) rc=$?"
fi
timeout -k 10 900 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "$RE" -d $O/w -o w --output-format csv -- python3 -X faulthandler $PMC_CMD > $O/w.log 2> $O/w.err
echo "[profile] WRITE_SIZE rc=$?"
python3 profiles/make_pmc_summary.py $O/f/f_counter_collection.csv $O/w/w_counter_collection.csv $PLAST $WL $O/f.log $COMMIT > $O/pmc_summary.json 2> $O/pmc_summary.err
timeout -k 10 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-include-regex "$RE" -d $O/sq -o sq --output-format csv -- python3 -X faulthandler $PMC_CMD > $O/sq.log 2> $O/sq.err
echo "[profile] SQ rc=$?"
python3 profiles/make_sq_summary.py $O/sq/sq_counter_collection.csv $PLAST $WL > $O/sq_summary.json 2> $O/sq_summary.err
if [ -n "$CRASH" ]; then
  ZT_DUMP_MAPS=$O/crash_maps.txt timeout -k 10 900 rocprofv3 --pmc FETCH_SIZE -d $O/fa -o fa --output-format csv -- python3 -X faulthandler $PMC_CMD > $O/fa.log 2> $O/crash_all_dispatches.err
  echo "[profile] FETCH_SIZE, every dispatch instrumented rc=$?"
  python3 tools/symbolise.py $O/crash_maps.txt $O/crash_all_dispatches.err > $O/crash_all_dispatches_symbolised.txt 2>&1
  rm -rf $O/fa
fi
ls -la $O/f $O/w $O/sq 2>/dev/null | head -20
rm -rf $O/f $O/w $O/sq $O/kt
for f in $O/*.err; do echo "== $f"; grep -v "simple_timer\|tool.cpp\|output_stream\|amdgpu.ids" $f | tail -8; done
head -16 $O/kernel_trace_summary.txt
