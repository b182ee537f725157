#!/bin/bash
# The rocprofv3 counter-mode crash (profiles/r5/rocprofv3_counter_crash/): the faulting frames go from torch's BUNDLED
# libamdhip64.so (ROCm 7.0, torch/lib, no SONAME version) into the /opt/rocm-7.2.0 libhsa-runtime64.so.1 that the profiler
# preloads, and from there into librocprofiler-sdk 1.1.0 -- two ROCm stacks in one process.  This runs the UNFILTERED
# FETCH_SIZE pass (every dispatch instrumented: the pass that crashes) with ONE stack: the system HIP preloaded, so that
# every hip* symbol of the process (torch's and libzebra_amd.so's) resolves to ROCm 7.2's runtime, whose HSA and
# rocprofiler-register are the profiler's own; then the control without the preload.
#   gpurun -- 'bash tools/exp/rocprof_one_stack.sh'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6/onestack; rm -rf $O; mkdir -p $O
CMD="bench.py --workload c5 --steps 20 --warmup 5 --legs none --cpu-edges 0 --no-score --steady-steps 0 --no-profile"
HIP=/opt/rocm/lib/libamdhip64.so.7
LD_PRELOAD=$HIP timeout -k 10 300 python3 $CMD > $O/plain_one_stack.json 2> $O/plain_one_stack.err
echo "[one-stack] bench alone under the system HIP: rc=$?"
LD_PRELOAD=$HIP ZT_DUMP_MAPS=$O/maps_one_stack.txt timeout -k 10 700 rocprofv3 --pmc FETCH_SIZE -d $O/fa -o fa --output-format csv -- python3 -X faulthandler $CMD > $O/fa_one_stack.log 2> $O/fa_one_stack.err
echo "[one-stack] FETCH_SIZE, every dispatch instrumented, ONE stack: rc=$?"
ls $O/fa 2>/dev/null | head -3; wc -l $O/fa/*counter_collection.csv 2>/dev/null | tail -1
grep -c "hsa-runtime64\|amdhip64" $O/maps_one_stack.txt 2>/dev/null
grep "hsa-runtime64\|amdhip64\|rocprofiler-register" $O/maps_one_stack.txt 2>/dev/null | awk '{print $NF}' | sort | uniq -c > $O/modules_one_stack.txt
rm -rf $O/fa
ZT_DUMP_MAPS=$O/maps_control.txt timeout -k 10 700 rocprofv3 --pmc FETCH_SIZE -d $O/fb -o fb --output-format csv -- python3 -X faulthandler $CMD > $O/fb_control.log 2> $O/fb_control.err
echo "[one-stack] control (torch's bundled HIP on the profiler's HSA): rc=$?"
grep "hsa-runtime64\|amdhip64\|rocprofiler-register" $O/maps_control.txt 2>/dev/null | awk '{print $NF}' | sort | uniq -c > $O/modules_control.txt
rm -rf $O/fb $O/maps_one_stack.txt $O/maps_control.txt
tail -3 $O/fa_one_stack.err | cut -c1-300; tail -3 $O/fb_control.err | cut -c1-300
cat $O/modules_one_stack.txt $O/modules_control.txt
