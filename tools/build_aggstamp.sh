#!/bin/bash
# diagnostic build with per-phase clocks in k_fc1_agg -> tools/out/libzebra_aggstamp.so
set -e
cd /root/repo/zebra_amd/csrc
O=/root/repo/tools/out
mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DZT_AGG_STAMP -c aggregate.hip -o $O/agg.o
L=/root/repo/zebra_amd/lib
hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libzebra_aggstamp.so $O/agg.o $L/tppr_stream.o $L/tppr_prune.o $L/aggregate_bwd.o $L/pipeline.o $L/memory_update.o $L/train_ops.o $L/attention.o $L/test_hooks.o
