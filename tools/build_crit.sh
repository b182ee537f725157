#!/bin/bash
# diagnostic build with four register-held clock readings per hub hop -> tools/out/libzebra_crit.so
set -e
cd /root/repo/zebra_amd/csrc
O=/root/repo/tools/out
mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DZT_CRIT -c tppr_stream.hip -o $O/ts_crit.o
L=/root/repo/zebra_amd/lib
hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libzebra_crit.so $O/ts_crit.o $L/tppr_prune.o $L/aggregate.o $L/aggregate_bwd.o $L/memory_update.o $L/train_ops.o $L/attention.o $L/pipeline.o $L/test_hooks.o
