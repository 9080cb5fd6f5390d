"""ctypes binding of libuw_hip.so (C ABI: include/uw_hip.h).

The library is the product path: if it is missing, every entry point raises -- there is no CPU fallback.
Build it with `python -c "import __graft_entry__ as g; g.build()"` or `whisper_amd/csrc/build.sh`.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libuw_hip.so")

UW_DTYPE_F32 = 0
UW_DTYPE_BF16 = 1
UW_MAX_ALIGN_HEADS = 32


class UwConfig(C.Structure):
    _fields_ = [
        ("d_model", C.c_int32), ("n_heads", C.c_int32), ("enc_layers", C.c_int32), ("dec_layers", C.c_int32),
        ("vocab", C.c_int32), ("n_mels", C.c_int32), ("n_audio_ctx", C.c_int32), ("n_text_ctx", C.c_int32),
        ("dtype", C.c_int32), ("max_batch", C.c_int32), ("n_align_heads", C.c_int32),
        ("align_layer", C.c_int32 * UW_MAX_ALIGN_HEADS), ("align_head", C.c_int32 * UW_MAX_ALIGN_HEADS),
    ]


class UwError(RuntimeError):
    pass


_vp, _i32, _i64 = C.c_void_p, C.c_int32, C.c_int64

# name -> (restype, argtypes); every symbol include/uw_hip.h declares
SIGNATURES = {
    "uw_last_error": (C.c_char_p, []),
    "uw_abi_version": (C.c_int, []),
    "uw_ctx_create": (C.c_int, [C.POINTER(UwConfig), C.POINTER(_vp), C.c_int, _vp, C.POINTER(_vp)]),
    "uw_ctx_destroy": (C.c_int, [_vp]),
    "uw_ctx_bytes": (_i64, [_vp]),
    "uw_mel": (C.c_int, [_vp, _vp, _i64, _vp, C.c_int, _vp, _vp]),
    "uw_encode": (C.c_int, [_vp, _vp, C.c_int, _vp, _vp]),
    "uw_cross_kv": (C.c_int, [_vp, _vp, C.c_int, _vp]),
    "uw_decoder_forward": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp]),
    "uw_decode_step": (C.c_int, [_vp, _vp, _vp, C.c_int, _vp, _vp]),
    "uw_decode_greedy": (C.c_int, [_vp, _vp, C.c_int, C.POINTER(_i32), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_int, _vp, _vp, _vp, _vp, _vp]),
    "uw_window_status": (C.c_int, [_vp, C.c_int, C.POINTER(_i32)]),
    "uw_attn_rows_forced": (C.c_int, [_vp, _vp, C.c_int, C.POINTER(_i32), _vp, C.c_int, C.POINTER(_i32), C.c_int, _vp]),
    "uw_get_attn_rows": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp]),
    "uw_cost_matrix": (C.c_int, [_vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "uw_dtw_workspace_bytes": (_i64, [C.c_int, C.c_int, C.c_int]),
    "uw_dtw": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp]),
    "uw_dtw_static": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp]),
    "uw_align": (C.c_int, [_vp, _vp, C.c_int, _vp, _vp]),
    "uw_transcribe_windows": (C.c_int, [_vp, _vp, _i64, _vp, _vp, C.c_int, C.POINTER(_i32), C.c_int, C.c_int, C.c_int,
                                        C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp]),
    "uw_profile_begin": (C.c_int, [_vp, C.c_int]),
    "uw_profile_end": (C.c_int, [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64), C.POINTER(C.c_double)]),
    "uw_profile_span": (C.c_int, [_vp, C.c_int, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(C.c_double)]),
}

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Loads the shared library once and types every entry point.  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise UwError(f"{LIB_PATH} not found: the HIP library must be built (whisper_amd/csrc/build.sh); "
                          "there is no CPU fallback")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc: int) -> None:
    if rc != 0:
        msg = load().uw_last_error()
        raise UwError(f"libuw_hip error {rc}: {msg.decode() if msg else '?'}")


def ptr(t) -> int:
    """Raw device pointer of a torch tensor (None -> NULL)."""
    return 0 if t is None else t.data_ptr()


def stream_handle() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream
