"""ctypes binding of libzebra_amd.so (the C-ABI declared in include/zebra_amd.h).

There is no CPU fallback: if the HIP library is missing or fails to load the
import raises, and every op raises on a non-zero status.
"""
import ctypes as C
import os

# torch first: it bundles its own ROCm runtime (libamdhip64); loading
# libzebra_amd.so afterwards makes both share that one HIP runtime instance, so
# torch device pointers and streams are valid inside the library.  Loaded the
# other way round the process ends up with two runtimes and hipMalloc fails
# with "no ROCm-capable device is detected".
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libzebra_amd.so")

ZT_OK = 0
ZT_ERR_ARG, ZT_ERR_RANGE, ZT_ERR_HIP, ZT_ERR_UNSUPPORTED, ZT_ERR_TIMEOUT = -1, -2, -3, -4, -5

# every symbol include/zebra_amd.h declares (tests check the library exports them all)
SYMBOLS = [
    "zt_last_error", "zt_version", "zt_set_kernel_choice", "zt_profile_enable", "zt_profile_reset", "zt_profile_read", "zt_stream_create_masked", "zt_stream_destroy",
    "zt_tppr_create", "zt_tppr_destroy", "zt_tppr_set_device_share", "zt_tppr_chain_stats", "zt_tppr_reset", "zt_tppr_copy", "zt_tppr_stream", "zt_tppr_plan", "zt_tppr_status",
    "zt_tppr_export", "zt_tppr_export_rows", "zt_tppr_import", "zt_tppr_import_rows",
    "zt_csr_build", "zt_csr_from_sorted", "zt_csr_size", "zt_csr_export", "zt_csr_destroy", "zt_csr_find_before", "zt_pruned_topk", "zt_pruned_topk_multi",
    "zt_embed_workspace_bytes", "zt_embed", "zt_project_table_bytes", "zt_project_memory", "zt_agg_train_forward", "zt_agg_backward_workspace_bytes", "zt_agg_train_backward", "zt_pipeline_create", "zt_pipeline_destroy", "zt_pipeline_main_stream", "zt_pipeline_update", "zt_pipeline_step", "zt_pipeline_step_ahead", "zt_pipeline_set_group",
    "zt_store_messages", "zt_store_messages_range", "zt_gru_workspace_bytes", "zt_gru_rows_offset", "zt_gru_update", "zt_gemm_f32", "zt_colsum_f32", "zt_overlay_rows", "zt_overlay_rows_backward", "zt_gru_train_workspace_bytes", "zt_gru_train_forward", "zt_gru_train_backward", "zt_pipeline_set_stats", "zt_pipeline_outstanding", "zt_pack_rows", "zt_scatter_rows", "zt_attention_workspace_bytes", "zt_temporal_attention",
    "zt_affinity_workspace_bytes", "zt_affinity", "zt_link_metrics", "zt_pipeline_set_scoring", "zt_pipeline_last_scores", "zt_pipeline_run",
    "zt_exchange_unique_id", "zt_exchange_create", "zt_exchange_set_tables", "zt_exchange_destroy", "zt_pipeline_set_exchange",
]


# zt_set_kernel_choice selectors / values (include/zebra_amd.h)
CHOICE_AGGREGATE, CHOICE_EMBED_OUT, CHOICE_GRU, CHOICE_MESSAGES, CHOICE_TPPR_CHAIN, CHOICE_TPPR_PREPASS = 0, 1, 2, 3, 4, 5
CHOICE_GROUP_RELEASE = 6
RELEASE_MEMBER, RELEASE_LAUNCH, RELEASE_LAUNCH_FULL = 1, 2, 3
PREPASS_LAUNCHES, PREPASS_COOP = 1, 2
CHAIN_SINGLE, CHAIN_PAIRED, CHAIN_SPINE, CHAIN_DUO = 1, 2, 3, 4
AGG_GENERIC = 1
OUT_TILED, OUT_LATENCY, OUT_PERSIST = 1, 2, 3
GRU_TILE, GRU_SPLIT = 1, 2
MSG_ONE, MSG_TWO = 1, 2


def set_kernel_choice(which, value):
    """Pin one of the kernels of a step process-wide (0: the library picks by shape again) -- for tests that hold the
    kernels against each other."""
    check(lib().zt_set_kernel_choice(C.c_int32(which), C.c_int32(value)), "zt_set_kernel_choice")
    _choices[which] = value


_choices = {}


def kernel_choice(which):
    """What set_kernel_choice last pinned for selector `which` in this process (0: the library's pick)."""
    return _choices.get(which, 0)


class RowTables(C.Structure):
    _fields_ = [("ptr", C.c_void_p * 8), ("width", C.c_int32 * 8), ("n", C.c_int32)]


class EmbedWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("fc1_w", "fc1_b", "fc2_w", "fc2_b", "fc1s_w", "fc1s_b", "fc2s_w",
                                          "fc2s_b", "time_w")]


class AttnWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("q_w", "k_w", "v_w", "in_b", "out_w", "out_b", "m1_w", "m1_b", "m2_w", "m2_b")]


class AffinityWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("fc1_w", "fc1_b", "fc2_w", "fc2_b")]


class GruWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("w_ih", "w_hh", "b_ih", "b_hh")]


class PipelineDesc(C.Structure):
    _fields_ = [("tppr", C.c_void_p), ("csr", C.c_void_p), ("width", C.c_int32), ("depth", C.c_int32),
                ("alpha", C.c_double * 16), ("beta", C.c_double * 16),
                ("memory", C.c_void_p), ("last_update", C.c_void_p), ("messages", C.c_void_p), ("msg_ts", C.c_void_p),
                ("flags", C.c_void_p), ("scratch", C.c_void_p), ("efeat", C.c_void_p),
                ("num_nodes", C.c_int64), ("num_edges", C.c_int64),
                ("D", C.c_int32), ("F", C.c_int32), ("T", C.c_int32), ("M", C.c_int32), ("k", C.c_int32),
                ("ew", EmbedWeights), ("gw", GruWeights),
                ("embed_ws", C.c_void_p), ("gru_ws", C.c_void_p), ("proj_table", C.c_void_p), ("status", C.c_void_p),
                ("max_B", C.c_int64)]


XCHG_RCCL, XCHG_SHM = 1, 2


class ExchangeDesc(C.Structure):
    _fields_ = [("rank", C.c_int32), ("world", C.c_int32), ("transport", C.c_int32), ("with_messages", C.c_int32),
                ("unique_id", C.c_void_p), ("shm_name", C.c_char_p), ("cap_rows", C.c_int64),
                ("memory", C.c_void_p), ("last_update", C.c_void_p), ("messages", C.c_void_p), ("msg_ts", C.c_void_p),
                ("D", C.c_int32), ("msg_dim", C.c_int32)]


class Batch(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("neg", C.c_void_p), ("ts", C.c_void_p), ("eidx", C.c_void_p),
                ("B", C.c_int64)]


_lib = None


def lib():
    """Load the HIP library (in-tree build).  Fails loudly when absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "zebra_amd: %s not found. Build it with `python -m zebra_amd.build` "
                "(hipcc, --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
        _lib = C.CDLL(LIB_PATH)
        _lib.zt_last_error.restype = C.c_char_p
        if hasattr(_lib, "zt_pipeline_main_stream"):
            _lib.zt_pipeline_main_stream.restype = C.c_void_p
        _lib.zt_version.restype = C.c_char_p
        for name in ("zt_embed_workspace_bytes", "zt_gru_workspace_bytes", "zt_gru_rows_offset", "zt_gru_train_workspace_bytes", "zt_attention_workspace_bytes", "zt_project_table_bytes", "zt_agg_backward_workspace_bytes", "zt_affinity_workspace_bytes"):
            if hasattr(_lib, name):
                getattr(_lib, name).restype = C.c_int64
    return _lib


_hooks = None


def hooks_lib():
    """libzebra_amd_testhooks.so: direct access to device primitives for the TESTS (zt_test_topk, zt_test_set_epoch;
    zebra_amd/csrc/test_hooks.h).  Not part of the product library; the product never loads it."""
    global _hooks
    if _hooks is None:
        lib()                                           # the hooks resolve their symbols against the product library
        path = os.path.join(os.path.dirname(LIB_PATH), "libzebra_amd_testhooks.so")
        if not os.path.exists(path):
            raise ImportError("zebra_amd: %s not found (python -m zebra_amd.build)" % path)
        _hooks = C.CDLL(path)
    return _hooks


class ZebraError(RuntimeError):
    pass


def check(rc, what=""):
    if rc == ZT_OK:
        return
    msg = lib().zt_last_error().decode()
    text = "%s: %s (status %d)" % (what or "zebra_amd", msg, rc)
    if rc == ZT_ERR_RANGE:
        raise IndexError(text)
    if rc in (ZT_ERR_ARG, ZT_ERR_UNSUPPORTED):
        raise ValueError(text)
    raise ZebraError(text)


def ptr(t):
    """Device/host address of a torch tensor or numpy array (None -> NULL)."""
    if t is None:
        return None
    if hasattr(t, "data_ptr"):
        return C.c_void_p(t.data_ptr())
    return t.ctypes.data_as(C.c_void_p)


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def to_device(device, arrays):
    """Several host arrays to the device with ONE transfer: packed into a pinned staging buffer (torch's caching host
    allocator: reused once the copy that read it has completed) and copied asynchronously on the current stream, so the host
    does not wait for the work already queued there -- eight pageable copies per training step were eight synchronisations
    (round 6: 1.3 of the step's 4.2 ms).  Returns device tensors of the arrays' dtypes and shapes."""
    import numpy as np
    import torch
    arrs = [np.ascontiguousarray(a) for a in arrays]
    offs, total = [], 0
    for a in arrs:
        offs.append(total)
        total += (a.nbytes + 15) & ~15
    host = torch.empty(max(total, 16), dtype=torch.uint8, pin_memory=True)
    hv = host.numpy()
    for a, o in zip(arrs, offs):
        hv[o:o + a.nbytes] = a.reshape(-1).view(np.uint8)
    dev = host.to(device, non_blocking=True)
    out = []
    for a, o in zip(arrs, offs):
        t = dev[o:o + a.nbytes].view(getattr(torch, str(a.dtype)))
        out.append(t.view(a.shape))
    return out
