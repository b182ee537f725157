"""zebra_amd -- MI355X (gfx950) implementation of Zebra's hot path:
streaming / pruning top-k T-PPR, top-k gather + aggregate, TGN memory update.

The compute lives in zebra_amd/lib/libzebra_amd.so (hand-written HIP, C-ABI in
include/zebra_amd.h).  The Python classes mirror the reference's surface.
"""
__version__ = "0.1.0"
