#!/bin/bash
# A DIAGNOSTIC copy of the library (-DZT_DIAG): the switches that give wrong results by construction -- ZT_AGG_DBG
# (k_fc1_agg_reg with parts removed: tools/exp/agg_variants.py), ZT_PRUNE_STOP (k_pruned_topk stopped after a phase:
# tools/exp/prune_phases.py) -- and zt_debug_regclk exist only here, never in zebra_amd/lib/libzebra_amd.so.
#   tools/build_diag.sh          builds zebra_amd/lib/ IN PLACE with -DZT_DIAG (run `python -m zebra_amd.build --force` afterwards
#                                to get the product library back); the tools load whatever zebra_amd/lib holds.
set -e
cd "$(dirname "$0")/.."
ZT_EXTRA_HIPFLAGS="-DZT_DIAG $ZT_EXTRA_HIPFLAGS" python3 -m zebra_amd.build --force
nm -D zebra_amd/lib/libzebra_amd.so | grep -c zt_debug_regclk
