#!/bin/bash
# diagnostic build with register-held clock readings per hub hop (-DZT_CRIT) -> tools/out/libzebra_crit.so
set -e
cd /root/repo/zebra_amd/csrc
O=/root/repo/tools/out
mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DZT_CRIT -c tppr_stream.hip -o $O/ts_crit.o
L=/root/repo/zebra_amd/lib
OBJS=$(ls $L/*.o | grep -v "tppr_stream.o\|test_hooks.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libzebra_crit.so $O/ts_crit.o $OBJS
