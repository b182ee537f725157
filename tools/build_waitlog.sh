#!/bin/bash
# diagnostic build with the per-task wait log (dumped by zt_tppr_status on a time-out) -> tools/out/libzebra_waitlog.so
set -e
cd /root/repo/zebra_amd/csrc
O=/root/repo/tools/out
mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DZT_WAITLOG -c tppr_stream.hip -o $O/ts_wl.o
L=/root/repo/zebra_amd/lib
OBJS=$(ls $L/*.o | grep -v "tppr_stream.o\|test_hooks.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libzebra_waitlog.so $O/ts_wl.o $OBJS
