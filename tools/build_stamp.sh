#!/bin/bash
# diagnostic build with in-kernel stamps -> tools/out/libzebra_stamp.so
set -e
cd /root/repo/zebra_amd/csrc
O=/root/repo/tools/out
mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DZT_STAMP -c tppr_stream.hip -o $O/ts.o
L=/root/repo/zebra_amd/lib
OBJS=$(ls $L/*.o | grep -v "tppr_stream.o\|test_hooks.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libzebra_stamp.so $O/ts.o $OBJS
