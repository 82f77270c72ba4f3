"""ctypes binding of libdecompdiff_hip.so (the C ABI of include/decompdiff_hip.h).

PyTorch is used only as the owner of device memory and streams: tensors are handed to the
library as raw device pointers (``tensor.data_ptr()``) and launches go to torch's current HIP
stream.  There is NO fallback: if the shared library is missing or fails to load, every entry
point raises — the product path never computes on the CPU.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_int64, c_long, c_size_t, c_uint64, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# DD_HIP_LIB selects another build of the same ABI (used by tools/ab_builds.py to compare two builds on one box)
LIB_PATH = os.environ.get("DD_HIP_LIB") or os.path.join(_HERE, "lib", "libdecompdiff_hip.so")

# the drop-in boundary: include/decompdiff_hip.h
EXPORTED_SYMBOLS = [
    "dd_status_string", "dd_abi_version", "dd_weights_form", "dd_build_flags", "dd_workspace_floats", "dd_knn", "dd_knn_masked", "dd_edge_weights", "dd_gemm128",
    "dd_gemm128_tn", "dd_gemm128_tn_bias", "dd_gemm128_tn_scratch_floats",
    "dd_ln_relu_scratch_floats", "dd_ln_relu_forward", "dd_ln_relu_backward",
    "dd_embed_protein", "dd_forward", "dd_sample_steps", "dd_sample_steps_graph", "dd_sample_steps_graph_multi",
    "dd_graph_create", "dd_graph_launch", "dd_graph_destroy",
    "dd_drift_armsca", "dd_drift_clash", "dd_drift_arms_repul",
    "dd_attn_aggregate_node", "dd_attn_aggregate_triplet", "dd_attn_aggregate_pos", "dd_reverse_step",
    "dd_segment_reduce", "dd_segment_softmax", "dd_sampler_reset",
    "dd_layer0_tables", "dd_layer0_prepare",
]
# measurement / profiling / test access: include/decompdiff_hip_debug.h (same library, not part of the boundary)
DEBUG_SYMBOLS = [
    "dd_workspace_view", "dd_profile_step", "dd_debug_set_clock_buffer", "dd_debug_set_fusion", "dd_debug_set_option",
    "dd_debug_node_split", "dd_debug_node_split_cache_path", "dd_debug_schedule", "dd_debug_philox", "dd_debug_options_epoch", "dd_queue_error",
]
# exported by the measurement build only (-DDD_DEBUG_OPTIONS=1): the tile-queue schedule's sticky error word
MEASUREMENT_ONLY_SYMBOLS = ("dd_queue_error",)
BUILD_DEBUG_OPTIONS, BUILD_EXACT_MATH = 1, 2          # bits of dd_build_flags()


class DDSampler(ctypes.Structure):
    """struct dd_sampler — field order must match include/decompdiff_hip.h exactly."""
    _fields_ = [
        ("B", c_int32), ("NP", c_int32), ("NL", c_int32), ("K", c_int32), ("NF", c_int32),
        ("num_layers", c_int32), ("T", c_int32), ("t_start", c_int32),
        ("weights", c_void_p), ("slot_off", c_void_p),
        ("tab_pos", c_void_p), ("tab_v", c_void_p), ("tab_b", c_void_p), ("tab_score", c_void_p),
        ("protein_pos", c_void_p), ("protein_h", c_void_p), ("lig_aux", c_void_p), ("atom_std", c_void_p),
        ("offset", c_void_p), ("decomp_index", c_void_p), ("full_protein_pos", c_void_p),
        ("lig_pos", c_void_p), ("lig_v", c_void_p), ("lig_bond", c_void_p), ("step_counter", c_void_p),
        ("drift_armsca", c_int32), ("armsca_min_d", c_float), ("armsca_max_d", c_float), ("armsca_scale", c_int32),
        ("drift_clash", c_int32), ("clash_sigma", c_float), ("clash_gamma", c_float), ("clash_scale", c_int32),
        ("drift_norm_batch", c_int32),
        ("u_v", c_void_p), ("u_b", c_void_p), ("eps", c_void_p), ("seed", c_uint64),
        ("traj_pos", c_void_p), ("traj_v", c_void_p), ("traj_bond", c_void_p), ("traj_v0", c_void_p),
        ("traj_vt", c_void_p), ("traj_bt", c_void_p),
        ("pred_pos", c_void_p), ("pred_v", c_void_p), ("pred_bond", c_void_p),
        ("workspace", c_void_p), ("workspace_floats", c_size_t),
        ("np_real", c_void_p), ("nl_real", c_void_p), ("bl_prefix", c_void_p),
        ("l0_tables", c_void_p), ("l0_P", c_void_p), ("l0_qn", c_void_p),
        ("num_v", c_int32), ("drift_repul", c_int32), ("repul_max_d", c_float), ("repul_scale", c_int32),
    ]


L0_TABLE_FLOATS = 16 * 640 + 16 * 1280 + 5 * 640 + 16 * 128 + 16 * 128 + 80 * 128      # DD_L0_TABLE_FLOATS


class DDWsView(ctypes.Structure):
    _fields_ = [("x", c_void_p), ("h", c_void_p), ("hb", c_void_p), ("ew", c_void_p), ("A", c_void_p),
                ("nbr", c_void_p), ("Anb", c_void_p), ("lin_in_node", ctypes.c_int32)]


PROF_CATS = ["misc", "gemm", "assemble", "attn_NE", "attn_NB", "attn_BL", "attn_PE", "attn_PB", "step", "event_pair"]


ABI_VERSION = 9          # include/decompdiff_hip.h: layout of struct dd_sampler and of the tables it points to


class HipLibraryError(RuntimeError):
    pass


_lib = None


def load():
    """Load the shared library (once).  Raises HipLibraryError if it is absent — build it with
    ``python -m decompdiff_amd.build`` (or ``__graft_entry__.build()``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(f"{LIB_PATH} not found: build it with `python -m decompdiff_amd.build`; "
                              "there is no CPU fallback for the sampling hot path")
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:                                   # e.g. libamdhip64 missing
        raise HipLibraryError(f"could not load {LIB_PATH}: {e}") from e
    lib.dd_status_string.restype = c_char_p
    lib.dd_status_string.argtypes = [c_int]
    lib.dd_abi_version.restype = c_int
    ignore_abi = os.environ.get("DD_IGNORE_ABI") == "1"                                    # (A/B timing of older builds)
    if lib.dd_abi_version() != ABI_VERSION and not ignore_abi:
        raise HipLibraryError(f"{LIB_PATH} has ABI version {lib.dd_abi_version()}, this package needs {ABI_VERSION}: rebuild it")
    S = POINTER(DDSampler)
    # name -> argtypes (restype c_int = status code unless listed in `restypes`)
    protos = {
        "dd_build_flags": [],
        "dd_weights_form": [],
        "dd_workspace_floats": [c_int, c_int, c_int, c_int],
        "dd_knn": [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p],
        "dd_knn_masked": [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p],
        "dd_edge_weights": [c_void_p, c_void_p, c_int, c_int, c_int] + [c_void_p] * 7,
        "dd_gemm128": [c_void_p, c_int, c_long, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_long, c_int, c_int,
                       c_int, c_void_p],
        "dd_gemm128_tn_scratch_floats": [c_long, c_int],
        "dd_gemm128_tn": [c_void_p, c_int, c_int, c_void_p, c_int, c_long, c_void_p, c_void_p, c_int, c_int, c_void_p],
        "dd_gemm128_tn_bias": [c_void_p, c_int, c_int, c_void_p, c_int, c_long, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p],
        "dd_ln_relu_scratch_floats": [c_long],
        "dd_ln_relu_forward": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_void_p],
        "dd_ln_relu_backward": [c_void_p] * 9 + [c_long, c_void_p],
        "dd_embed_protein": [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p],
        "dd_forward": [S, c_void_p],
        "dd_sampler_reset": [S, c_void_p],
        "dd_layer0_tables": [S, c_void_p, c_void_p],
        "dd_layer0_prepare": [S, c_void_p],
        "dd_debug_options_epoch": [],
        "dd_queue_error": [S, c_void_p, POINTER(c_int)],
        "dd_sample_steps": [S, c_int, c_void_p],
        "dd_sample_steps_graph": [S, c_int, c_void_p],
        "dd_graph_create": [S, c_int, c_void_p, POINTER(c_void_p)],
        "dd_graph_launch": [c_void_p, c_int, c_void_p],
        "dd_graph_destroy": [c_void_p],
        "dd_sample_steps_graph_multi": [POINTER(S), c_int, c_int, POINTER(c_void_p)],
        "dd_drift_armsca": [c_void_p, c_void_p, c_int, c_int, c_float, c_float, c_void_p, c_int, c_void_p],
        "dd_drift_clash": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float, c_void_p, c_int, c_void_p],
        "dd_drift_arms_repul": [c_void_p, c_void_p, c_int, c_int, c_float, c_int, c_void_p, c_int, c_void_p],
        "dd_workspace_view": [S, POINTER(DDWsView)],
        "dd_debug_set_clock_buffer": [c_void_p, c_int],
        "dd_debug_set_fusion": [c_int],
        "dd_debug_set_option": [c_int, c_int],
        "dd_debug_node_split": [c_int, c_int, c_int, c_int],
        "dd_debug_node_split_cache_path": [c_char_p, c_int],
        "dd_debug_schedule": [],
        "dd_reverse_step": [S, c_void_p, c_void_p, c_void_p, c_void_p],
        "dd_attn_aggregate_node": [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p],
        "dd_attn_aggregate_triplet": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p],
        "dd_attn_aggregate_pos": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p],
        "dd_segment_reduce": [c_void_p, c_void_p, c_int, c_int, c_int, c_long, c_void_p, c_void_p, c_void_p],
        "dd_segment_softmax": [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p],
        "dd_debug_philox": [c_uint64, c_int, c_long, c_int, c_void_p, c_void_p],
        "dd_profile_step": [S, c_int, POINTER(c_float), c_void_p],
    }
    restypes = {"dd_workspace_floats": c_size_t, "dd_gemm128_tn_scratch_floats": c_size_t, "dd_ln_relu_scratch_floats": c_size_t}
    for name in EXPORTED_SYMBOLS + DEBUG_SYMBOLS:
        if name in ("dd_status_string", "dd_abi_version"):
            continue
        if not hasattr(lib, name):
            # DD_IGNORE_ABI=1 with an older build (tools/ab_builds.py): entry points it lacks raise when CALLED, not at load
            if ignore_abi or name in MEASUREMENT_ONLY_SYMBOLS:
                continue
            raise HipLibraryError(f"{LIB_PATH} does not export {name}: rebuild it")
        fn = getattr(lib, name)
        if name in protos:
            fn.argtypes = protos[name]
        fn.restype = restypes.get(name, c_int)
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().dd_status_string(rc).decode()
        raise RuntimeError(f"decompdiff_hip {what} failed: {msg} (status {rc})")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_contiguous(), "decompdiff_hip needs contiguous tensors"
    return ctypes.c_void_p(t.data_ptr())


def stream_ptr(device=None):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_gpu(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise HipLibraryError(f"{name} must live on a HIP device (got {t.device}); the sampling hot path has no "
                              "CPU implementation in this package")


def weights_form() -> int:
    """Form of the packed attention MLPs the loaded library expects (include/decompdiff_hip.h: dd_weights_form; 0 for builds
    older than ABI 9, loaded with DD_IGNORE_ABI=1 by the A/B tools)."""
    lib = load()
    return int(lib.dd_weights_form()) if hasattr(lib, "dd_weights_form") else 0


def is_measurement_build() -> bool:
    """The loaded library was compiled with -DDD_DEBUG_OPTIONS=1 (lib/libdecompdiff_hip_dbg.so: alternative launch
    schedules behind dd_debug_set_option)."""
    lib = load()
    if not hasattr(lib, "dd_build_flags"):               # (an older build loaded with DD_IGNORE_ABI=1)
        return False
    return bool(lib.dd_build_flags() & BUILD_DEBUG_OPTIONS)
