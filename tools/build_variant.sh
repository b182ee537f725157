#!/bin/bash
# a variant of the library that differs in tppr_stream.hip's compile flags only -> tools/out/libzebra_NAME.so
#   tools/build_variant.sh NAME "-DZT_SOMETHING=1 ..."      (run it with tools/exp/bench_lib.py tools/out/libzebra_NAME.so <bench.py args>)
set -e
NAME=${1:?name}; FLAGS=$2
cd /root/repo/zebra_amd/csrc
O=/root/repo/tools/out
mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $FLAGS -c tppr_stream.hip -o $O/ts_$NAME.o
L=/root/repo/zebra_amd/lib
OBJS=$(ls $L/*.o | grep -v "tppr_stream.o\|test_hooks.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libzebra_$NAME.so $O/ts_$NAME.o $OBJS -ldl -lrt
echo $O/libzebra_$NAME.so
