"""Builds zebra_amd/lib/libzebra_amd.so (HIP, gfx950) in-tree with hipcc.

    python -m zebra_amd.build [--force]

The T-PPR translation units are compiled with -ffp-contract=off: the reference
rounds every float64 multiply and add separately and the results must be
bit-exact.  hipcc cross-compiles without a GPU.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libzebra_amd.so")

# file -> extra flags
SOURCES = {
    "runtime.hip": [],
    # (k_reserve's atomicAdd of a per-lane count: the compiler's default wraps it in a scan over the lanes one at a time, ~580
    #  scalar instructions per wave; the DPP form of the same scan is ~20)
    "tppr_prepass.hip": ["-mllvm", "-amdgpu-atomic-optimizer-strategy=DPP"],
    "tppr_stream.hip": ["-ffp-contract=off"],
    "tppr_io.hip": [],
    "tppr_prune.hip": ["-ffp-contract=off"],
    "aggregate.hip": [],
    "aggregate_wide.hip": [],
    "aggregate_bwd.hip": [],
    "memory_update.hip": [],
    "train_ops.hip": [],
    "attention.hip": [],
    "scoring.hip": [],
    "exchange.hip": [],
    "pipeline.hip": [],
}
# NOT part of the product library: direct access to device primitives for the tests (tests load it beside the library)
HOOK_SOURCES = {"test_hooks.hip": ["-ffp-contract=off"]}
HOOKS_LIB = os.path.join(LIBDIR, "libzebra_amd_testhooks.so")
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def csrc_sha16():
    """sha256 (first 16 hex digits) over the kernel sources (csrc/*.hip, *.hpp, *.h, names and contents): what a profile
    summary under profiles/ was measured on, and what bench.py holds it against before it quotes the summary's traffic."""
    import hashlib
    hsh = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".hpp", ".h")):
            hsh.update(f.encode())
            with open(os.path.join(CSRC, f), "rb") as fh:
                hsh.update(fh.read())
    return hsh.hexdigest()[:16]


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(LIBDIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "zebra_amd.h"))
    objs = []
    rebuilt = False
    for src, extra in SOURCES.items():
        sp = os.path.join(CSRC, src)
        if not os.path.exists(sp):
            continue
        op = os.path.join(LIBDIR, src.replace(".hip", ".o"))
        if force or _newer(sp, op) or any(_newer(h, op) for h in headers):
            cmd = [hipcc] + COMMON + extra + os.environ.get("ZT_EXTRA_HIPFLAGS", "").split() + ["-c", sp, "-o", op]
            if verbose:
                print(" ".join(cmd))
            subprocess.run(cmd, check=True)
            rebuilt = True
        objs.append(op)
    if rebuilt or not os.path.exists(LIB):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl", "-lrt"]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    # the test hooks: their own shared object, resolved against the product library next to it
    hobjs, hrebuilt = [], False
    for src, extra in HOOK_SOURCES.items():
        sp = os.path.join(CSRC, src)
        op = os.path.join(LIBDIR, src.replace(".hip", ".o"))
        if force or _newer(sp, op) or any(_newer(h, op) for h in headers):
            subprocess.run([hipcc] + COMMON + extra + ["-c", sp, "-o", op], check=True)
            hrebuilt = True
        hobjs.append(op)
    if hrebuilt or rebuilt or not os.path.exists(HOOKS_LIB):
        subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", HOOKS_LIB] + hobjs +
                       ["-L" + LIBDIR, "-lzebra_amd", "-Wl,-rpath,$ORIGIN"], check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
