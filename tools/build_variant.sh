#!/bin/bash
# a variant of the library that differs in ONE source file's compile flags -> tools/out/libzebra_NAME.so
#   tools/build_variant.sh NAME "-DZT_SOMETHING=1 ..." [source.hip, default tppr_stream.hip]
#   (run it with tools/exp/bench_lib.py tools/out/libzebra_NAME.so <bench.py args>)
# The alternatives that were measured slower than the library's pick are variant builds too (their sources: tools/exp/variants/):
#   tools/build_variant.sh chainvariants "-DZT_CHAIN_VARIANTS"                            (paired / spine / duo hub chains)
#   tools/build_variant.sh prepasscoop "-DZT_PREPASS_COOP_VARIANT -mllvm -amdgpu-atomic-optimizer-strategy=DPP" tppr_prepass.hip   (the prepass as one cooperative kernel; the second flag is what zebra_amd/build.py compiles that file with)
#   ZT_TEST_LIB=tools/out/chainvariants/libzebra_amd.so python -m pytest tests/test_tppr_gpu.py -m gpu -k "paired or spine or duo"
set -e
NAME=${1:?name}; FLAGS=$2; SRC=${3:-tppr_stream.hip}
cd /root/repo/zebra_amd/csrc
O=/root/repo/tools/out
mkdir -p $O
FP=""; case $SRC in tppr_stream.hip|tppr_prune.hip) FP="-ffp-contract=off";; esac
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $FP $FLAGS -I/root/repo/zebra_amd/csrc -I/root/repo/tools/exp/variants -c $SRC -o $O/v_$NAME.o
L=/root/repo/zebra_amd/lib
OBJS=$(ls $L/*.o | grep -v "/${SRC%.hip}.o\|test_hooks.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libzebra_$NAME.so $O/v_$NAME.o $OBJS -ldl -lrt
# the same library under the product's file name in a directory of its own, with the test hooks linked against it: what
# ZT_TEST_LIB=tools/out/NAME/libzebra_amd.so (tests/conftest.py) loads -- the hooks resolve against THIS build
mkdir -p $O/$NAME
cp $O/libzebra_$NAME.so $O/$NAME/libzebra_amd.so
hipcc --offload-arch=gfx950 -shared -fPIC -o $O/$NAME/libzebra_amd_testhooks.so $L/test_hooks.o -L$O/$NAME -lzebra_amd -Wl,-rpath,'$ORIGIN'
echo $O/libzebra_$NAME.so
