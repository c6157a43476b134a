"""ctypes binding of libpfnl_hip.so (include/pfnl_hip.h).

There is NO CPU fallback: if the shared library is missing or no gfx950 device is visible, every
compute entry point raises.  torch is imported *before* the library is loaded so that both resolve
the same ``libamdhip64.so.7`` (torch bundles one with that soname) — a torch-ROCm tensor's
``data_ptr()`` and ``torch.cuda.current_stream().cuda_stream`` can then be handed straight to the
C-ABI.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PFNL_HIP_LIB", os.path.join(_HERE, "lib", "libpfnl_hip.so"))

K_NAMES = ("nl_pack", "nl_attn", "conv0", "conv3x3", "conv1x1", "merge1", "tail")


class PFNLHipError(RuntimeError):
    """A libpfnl_hip call returned a negative status."""


class PFNLHipMissing(ImportError):
    """libpfnl_hip.so has not been built (run `python -c "import __graft_entry__ as g; g.build()"`)."""


class pfnl_config(C.Structure):
    _fields_ = [("num_frames", C.c_int32), ("scale", C.c_int32), ("mf", C.c_int32),
                ("num_block", C.c_int32), ("device_id", C.c_int32), ("reserved", C.c_int32 * 3)]


# name -> (restype, argtypes); exactly the symbols include/pfnl_hip.h declares
_vp, _i, _fp = C.c_void_p, C.c_int, C.POINTER(C.c_float)
SIGNATURES = {
    "pfnl_last_error": (C.c_char_p, []),
    "pfnl_version": (_i, []),
    "pfnl_device_count": (_i, [C.POINTER(_i)]),
    "pfnl_create": (_i, [C.POINTER(pfnl_config), C.POINTER(_vp)]),
    "pfnl_destroy": (_i, [_vp]),
    "pfnl_set_weight": (_i, [_vp, C.c_char_p, _vp, C.POINTER(C.c_int64), _i]),
    "pfnl_missing_weights": (_i, [_vp, C.POINTER(_i)]),
    "pfnl_finalize_weights": (_i, [_vp]),
    "pfnl_set_option": (_i, [_vp, C.c_char_p, C.c_char_p]),
    "pfnl_forward": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _i, _vp]),
    "pfnl_forward_strip": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "pfnl_workspace_bytes": (_i, [_vp, _i, _i, _i, C.POINTER(C.c_size_t)]),
    "pfnl_plan": (_i, [_vp, _i, _i, _i, C.c_char_p, C.c_size_t]),
    "pfnl_get_option": (_i, [_vp, C.c_char_p, C.c_char_p, C.c_size_t]),
    "pfnl_sync": (_i, [_vp]),
    "pfnl_range_reruns": (_i, [_vp, C.POINTER(C.c_longlong)]),
    "pfnl_range_flag": (_i, [_vp, C.POINTER(_i)]),
    "pfnl_host_alloc": (_i, [C.c_size_t, C.POINTER(_vp)]),
    "pfnl_host_free": (_i, [_vp]),
    "pfnl_profile_enable": (_i, [_vp, _i]),
    "pfnl_profile_reset": (_i, [_vp]),
    "pfnl_profile_read": (_i, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "pfnl_debug_tap": (_i, [_vp, C.c_char_p, _vp, C.c_size_t]),
    "pfnl_op_conv2d": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "pfnl_op_conv2_grouped": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "pfnl_op_conv3x3_accum": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "pfnl_op_conv3x3_accum_split16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "pfnl_op_conv1x1_split16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "pfnl_op_conv1x1_stream": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "pfnl_op_conv3x3_bf16": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp]),
    "pfnl_op_conv1x1_bf16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "pfnl_op_conv3x3_accum_bf16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "pfnl_op_conv1_conv10_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "pfnl_op_conv3x3_winograd": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp]),
    "pfnl_op_conv3x3_split16": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp]),
    "pfnl_op_conv3x3_split16_sf": (_i, [_i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp]),
    "pfnl_op_conv_small": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "pfnl_op_conv_small_pf_block": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "pfnl_op_conv1x1_split16_sf": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "pfnl_op_conv3x3_winograd_ws": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp]),
    "pfnl_op_nonlocal": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "pfnl_op_nonlocal_split16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "pfnl_op_nonlocal_f16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "pfnl_op_bicubic": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "pfnl_op_blur_decimate": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "pfnl_selftest_mfma": (_i, [_i]),
    "pfnl_op_nonlocal_embedded": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "pfnl_op_conv1_conv10_split16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "pfnl_op_conv1_conv10_split16_sf0": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "pfnl_op_conv2_chain_sf0": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "pfnl_op_nonlocal_block": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _i, _i, _i, _i, _vp]),
    "pfnl_op_conv0": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "pfnl_op_tail": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "pfnl_op_gather_windows": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "pfnl_op_quantise_u8": (_i, [_vp, _vp, C.c_size_t, _vp]),
    "pfnl_comm_get_unique_id": (_i, [_vp]),
    "pfnl_comm_init_rank": (_i, [_i, _i, _vp, _i, C.POINTER(_vp)]),
    "pfnl_comm_init_all": (_i, [_i, C.POINTER(_i), C.POINTER(_vp)]),
    "pfnl_comm_destroy": (_i, [_vp]),
    "pfnl_comm_rank": (_i, [_vp, C.POINTER(_i), C.POINTER(_i)]),
    "pfnl_comm_bcast_weights": (_i, [_vp, _vp, _i]),
    "pfnl_copy_weights": (_i, [_vp, _vp]),
    "pfnl_comm_bcast": (_i, [_vp, _vp, C.c_size_t, _i]),
    "pfnl_comm_allreduce_f64": (_i, [_vp, C.POINTER(C.c_double), _i, _i]),
    "pfnl_comm_barrier": (_i, [_vp]),
    "pfnl_comm_allgather": (_i, [_vp, _vp, _vp, C.c_size_t, _vp]),
}
COMM_ID_BYTES = 128
COMM_SUM, COMM_MAX = 0, 1

_lib: Optional[C.CDLL] = None


def load_library(path: str = LIB_PATH) -> C.CDLL:
    """Load (once) and type the library.  Raises PFNLHipMissing if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise PFNLHipMissing(f"{path} not found: build it with __graft_entry__.build() "
                             f"(make -C pfnl_amd/csrc); there is no CPU fallback")
    try:  # share torch's HIP runtime (same soname) when torch is present
        import torch  # noqa: F401
        # ... and, for pfnl_comm_*, torch's RCCL build (it is linked against that same runtime); comm.hip dlopens it lazily
        cand = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        if os.path.exists(cand):
            os.environ.setdefault("PFNL_RCCL_LIB", cand)
    except Exception:  # pragma: no cover - torch is plumbing, not a requirement of the C-ABI
        pass
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError if the library does not export the symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status: int) -> None:
    if status != 0:
        msg = load_library().pfnl_last_error()
        raise PFNLHipError(f"libpfnl_hip status {status}: {msg.decode() if msg else '?'}")


def device_count() -> int:
    n = C.c_int(0)
    check(load_library().pfnl_device_count(C.byref(n)))
    return n.value
