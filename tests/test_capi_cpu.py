"""CPU-side checks of the product library: it loads, exports every symbol the
header declares, and the host-only entry points behave (no GPU compute)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def capi():
    from zebra_amd import build
    build.build()
    from zebra_amd import _capi
    return _capi


def test_library_exports_every_declared_symbol(capi):
    hdr = open(os.path.join(ROOT, "include", "zebra_amd.h")).read()
    declared = set(re.findall(r"\b(zt_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(capi.SYMBOLS), declared ^ set(capi.SYMBOLS)
    lib = capi.lib()
    for s in declared:
        assert hasattr(lib, s), "libzebra_amd.so does not export %s" % s
    assert b"gfx950" in lib.zt_version()


def test_bad_arguments_fail_without_gpu(capi):
    lib = capi.lib()
    h = C.c_void_p()
    a = np.array([0.1]); b = np.array([0.9])
    assert lib.zt_tppr_create(C.byref(h), C.c_int64(0), C.c_int32(5), C.c_int32(1), capi.ptr(a), capi.ptr(b)) == capi.ZT_ERR_ARG
    assert lib.zt_tppr_create(C.byref(h), C.c_int64(10), C.c_int32(256), C.c_int32(1), capi.ptr(a), capi.ptr(b)) == capi.ZT_ERR_UNSUPPORTED
    assert b"k=256" in lib.zt_last_error()
    with pytest.raises(ValueError):
        capi.check(capi.ZT_ERR_UNSUPPORTED)
    with pytest.raises(IndexError):
        capi.check(capi.ZT_ERR_RANGE)


def test_round5_entry_points_validate_their_arguments(capi):
    """zt_set_kernel_choice, zt_tppr_set_device_share, zt_exchange_*: bad arguments are refused before any device call
    (the selector range and the descriptor are checked on the host)."""
    lib = capi.lib()
    assert lib.zt_set_kernel_choice(C.c_int32(99), C.c_int32(0)) == capi.ZT_ERR_ARG
    assert lib.zt_set_kernel_choice(C.c_int32(capi.CHOICE_GRU), C.c_int32(-1)) == capi.ZT_ERR_ARG
    assert lib.zt_set_kernel_choice(C.c_int32(capi.CHOICE_GRU), C.c_int32(capi.GRU_SPLIT)) == capi.ZT_OK
    assert lib.zt_set_kernel_choice(C.c_int32(capi.CHOICE_GRU), C.c_int32(0)) == capi.ZT_OK
    assert lib.zt_tppr_set_device_share(None, C.c_int32(2)) == capi.ZT_ERR_ARG
    x = C.c_void_p()
    d = capi.ExchangeDesc()
    assert lib.zt_exchange_create(C.byref(x), None) == capi.ZT_ERR_ARG
    d.rank, d.world, d.transport, d.cap_rows = 2, 2, capi.XCHG_RCCL, 16          # rank out of range, no tables
    assert lib.zt_exchange_create(C.byref(x), C.byref(d)) == capi.ZT_ERR_ARG
    assert lib.zt_exchange_unique_id(None, C.c_int64(128)) == capi.ZT_ERR_ARG
    buf = (C.c_char * 64)()
    assert lib.zt_exchange_unique_id(buf, C.c_int64(64)) == capi.ZT_ERR_ARG       # the id is 128 bytes
    assert lib.zt_pipeline_set_exchange(None, None) == capi.ZT_ERR_ARG
    assert lib.zt_exchange_destroy(None) == capi.ZT_OK


def test_k_stream_leaves_room_for_the_message_kernels(capi, tmp_path):
    """k_stream's workgroups sit two waves to a SIMD on the T-PPR stream's compute units and the message kernels of the
    step (k_last_pos, k_build_messages2: <= 64 registers) run BESIDE them: that needs 2 x (k_stream's registers, in
    granules of 8) + 64 <= 512, i.e. k_stream <= 224 vector registers.  At 225 (round 6, a replay-server experiment) the message
    kernels waited for every k_stream to end and the step DOUBLED; at 256 (round 5, the paired hop inlined) likewise.  The
    count is read from the built object's code-object metadata (no GPU needed)."""
    import re
    import subprocess
    obj = os.path.join(os.path.dirname(capi.LIB_PATH), "tppr_stream.o")
    llvm = "/opt/rocm/lib/llvm/bin"
    if not (os.path.exists(obj) and os.path.exists(os.path.join(llvm, "clang-offload-bundler"))):
        pytest.skip("no built object / no LLVM tools here (the GPU box runs the prebuilt library)")
    fat, co = str(tmp_path / "fat.bin"), str(tmp_path / "ts.co")
    subprocess.run([os.path.join(llvm, "llvm-objcopy"), "--dump-section=.hip_fatbin=" + fat, obj, str(tmp_path / "unused.o")], check=True)
    subprocess.run([os.path.join(llvm, "clang-offload-bundler"), "--unbundle", "--type=o",
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + co], check=True)
    notes = subprocess.run([os.path.join(llvm, "llvm-readelf"), "--notes", co], check=True, capture_output=True, text=True).stdout
    m = re.search(r"\.name:\s+\S*k_streamILi0E\S*.*?\.vgpr_count:\s+(\d+)", notes, re.S)
    assert m, "k_stream<0> not found in the code object's metadata"
    vgprs = int(m.group(1))
    assert vgprs <= 224, "k_stream<0> needs %d vector registers: the message kernels no longer fit beside it" % vgprs
