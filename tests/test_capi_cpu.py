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
    assert lib.zt_tppr_create(C.byref(h), C.c_int64(10), C.c_int32(64), C.c_int32(1), capi.ptr(a), capi.ptr(b)) == capi.ZT_ERR_UNSUPPORTED
    assert b"k=64" in lib.zt_last_error()
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
