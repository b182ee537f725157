#!/bin/bash
# diagnostic build with in-kernel stamps -> tools/out/libzebra_stamp.so
set -e
cd /root/repo/zebra_amd/csrc
O=/root/repo/tools/out
mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DZT_STAMP -c tppr_stream.hip -o $O/ts.o
hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libzebra_stamp.so $O/ts.o /root/repo/zebra_amd/lib/tppr_prune.o /root/repo/zebra_amd/lib/aggregate.o /root/repo/zebra_amd/lib/aggregate_bwd.o /root/repo/zebra_amd/lib/memory_update.o /root/repo/zebra_amd/lib/train_ops.o /root/repo/zebra_amd/lib/attention.o /root/repo/zebra_amd/lib/pipeline.o /root/repo/zebra_amd/lib/test_hooks.o
