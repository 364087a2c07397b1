"""ctypes binding of ``libideas_hip.so`` (C ABI in ``include/ideas_hip.h``).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C ideas_amd/csrc``.  There is NO
fallback: if the shared object is missing or a symbol is absent, importing an op raises immediately.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IDEAS_HIP_LIB", os.path.join(_HERE, "libideas_hip.so"))  # override only for A/B kernel experiments

NCHW, NHWC = 0, 1
F32 = 0
F32_B3 = 1   # f32 tensors, split-bf16 contraction (IDEAS_F32_B3)
BF16 = 2     # bf16 activations, bf16 MFMA with f32 accumulation, f32 master weights (IDEAS_BF16)


def act_dtype(t: torch.Tensor) -> int:
    """dtype enum of the elementwise / reduction entry points for an activation tensor."""
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise RuntimeError(f"ideas_amd: only float32 and bfloat16 activations are implemented, got {t.dtype}")


class ConvParams(C.Structure):
    """Mirror of ``ideas_conv_params`` (include/ideas_hip.h)."""
    _fields_ = [
        ("B", C.c_int), ("IH", C.c_int), ("IW", C.c_int), ("Cin", C.c_int),
        ("YH", C.c_int), ("YW", C.c_int), ("Cout", C.c_int),
        ("OH", C.c_int), ("OW", C.c_int),
        ("TY", C.c_int), ("TX", C.c_int),
        ("sy", C.c_int), ("sx", C.c_int), ("dy", C.c_int), ("dx", C.c_int), ("offy", C.c_int), ("offx", C.c_int),
        ("osy", C.c_int), ("osx", C.c_int), ("ooy", C.c_int), ("oox", C.c_int),
        ("reflect", C.c_int), ("act", C.c_int),
        ("alpha", C.c_float), ("act_gain", C.c_float), ("resid_gain", C.c_float),
        ("accumulate", C.c_int), ("gain", C.c_float),
    ]


class PrepDesc(C.Structure):
    """Mirror of ``ideas_prep_desc`` (include/ideas_hip.h): one parameter of a batched weight-preparation launch."""
    _fields_ = [("dst", C.c_void_p), ("w", C.c_void_p), ("s", C.c_int64 * 5), ("a", C.c_int * 4), ("unit", C.c_int),
                ("block0", C.c_int), ("nblocks", C.c_int), ("pad_", C.c_int)]


PREP_B3_SPLIT, PREP_B3_WINO, PREP_BF16_PACK = 0, 1, 2


class LinearSeg(C.Structure):
    """Mirror of ``ideas_linear_seg`` (include/ideas_hip.h): one layer of a batched EqualLinear launch."""
    _fields_ = [("w", C.c_void_p), ("bias", C.c_void_p), ("y", C.c_void_p), ("gw", C.c_void_p), ("gb", C.c_void_p),
                ("n", C.c_int), ("ldw", C.c_int), ("ldy", C.c_int), ("ldgw", C.c_int), ("scale", C.c_float), ("bias_mul", C.c_float),
                ("tile0", C.c_int), ("pad_", C.c_int)]


LINEAR_MAX_SEGMENTS = 32

_P = C.c_void_p
_PROTOS = {
    "ideas_abi_version": (C.c_int, []),
    "ideas_sizeof_conv_params": (C.c_int, []),
    "ideas_strerror": (C.c_char_p, [C.c_int]),
    "ideas_stream_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "ideas_stream_destroy": (C.c_int, [_P]),
    "ideas_fused_bias_act": (C.c_int, [_P, _P, _P, _P, _P, C.c_int64, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int,
                                       C.c_float, C.c_float, C.c_int, _P]),
    "ideas_upfirdn2d": (C.c_int, [_P, _P, _P] + [C.c_int] * 14 + [C.c_float, C.c_int, C.c_int, C.c_int, _P]),
    "ideas_fir_up2_add": (C.c_int, [_P, _P, _P, _P] + [C.c_int] * 8 + [C.c_float, C.c_int, C.c_int, _P]),
    "ideas_blur_fused": (C.c_int, [_P, _P, _P] + [C.c_int] * 8 + [C.c_float, C.c_int, C.c_int, _P, _P, _P, C.c_float, C.c_float,
                                   C.c_int, _P]),
    "ideas_conv_igemm": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.POINTER(ConvParams), C.c_int, _P]),
    "ideas_conv_igemm_multi": (C.c_int, [C.c_int, _P, _P, C.POINTER(C.c_void_p), _P, _P, C.POINTER(ConvParams), C.c_int, _P]),
    "ideas_b3_conv_supported": (C.c_int, [C.POINTER(ConvParams)]),
    "ideas_b3_wgrad_supported": (C.c_int, [C.POINTER(ConvParams)]),
    "ideas_b3_wgrad3_supported": (C.c_int, [C.POINTER(ConvParams)]),
    "ideas_b3_wino_supported": (C.c_int, [C.POINTER(ConvParams)]),
    "ideas_b3_wino_split_weights": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _P]),
    "ideas_b3_split_weights": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P]),
    "ideas_b3_split_weights_strided": (C.c_int, [_P, _P] + [C.c_int] * 4 + [C.c_int64] * 4 + [_P]),
    "ideas_bf16_conv_supported": (C.c_int, [C.POINTER(ConvParams), C.c_int]),
    "ideas_bf16_wgrad_supported": (C.c_int, [C.POINTER(ConvParams), C.c_int]),
    "ideas_bf16_direct_supported": (C.c_int, [C.POINTER(ConvParams)]),
    "ideas_bf16_pack_weights": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "ideas_bf16_pack_weights_strided": (C.c_int, [_P, _P, _P] + [C.c_int] * 5 + [C.c_int64] * 4 + [_P]),
    "ideas_conv3x3_wino": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.POINTER(ConvParams), C.c_int, _P]),
    "ideas_conv3x3_wino_wgrad": (C.c_int, [_P, _P, _P, _P, _P, C.POINTER(ConvParams), C.c_int, _P]),
    "ideas_wino_wgrad_fold": (C.c_int, [_P, _P, C.c_int, C.c_int] + [C.c_int64] * 4 + [C.c_int, _P]),
    "ideas_b3_blur_conv_s2_supported": (C.c_int, [C.POINTER(ConvParams), C.c_int, C.c_int, C.c_int]),
    "ideas_b3_blur_conv_s2": (C.c_int, [_P, _P, _P, _P, C.POINTER(C.c_float), C.POINTER(C.c_float), _P, _P, C.POINTER(ConvParams),
                                        C.c_int, C.c_int, C.c_int, _P]),
    "ideas_conv_direct": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.POINTER(ConvParams), C.c_int, _P]),
    "ideas_conv_wgrad": (C.c_int, [_P, _P, _P, _P, _P, C.POINTER(ConvParams), C.c_int, _P]),
    "ideas_conv_wgrad_direct": (C.c_int, [_P, _P, _P, _P, _P, C.POINTER(ConvParams), C.c_int, _P]),
    "ideas_demod": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_float, _P]),
    "ideas_weight_sqsum": (C.c_int, [_P, _P] + [C.c_int] * 4 + [C.c_int64] * 4 + [C.c_float, _P]),
    "ideas_weight_sqsum_f64": (C.c_int, [_P, _P] + [C.c_int] * 4 + [C.c_int64] * 4 + [C.c_double, _P]),
    "ideas_demod_bwd": (C.c_int, [_P] * 7 + [C.c_int] * 3 + [C.c_float, _P]),
    "ideas_demod_wgrad": (C.c_int, [_P] * 4 + [C.c_int] * 5 + [C.c_int64] * 8 + [C.c_float, _P]),
    "ideas_pixel_dot": (C.c_int, [_P, _P, _P, C.c_int, C.c_int64, C.c_int, C.c_int, _P]),
    "ideas_reflect_fold": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "ideas_adam_ema": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _P]),
    "ideas_image_u8_to_f32": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, _P]),
    "ideas_channel_sum": (C.c_int, [_P, _P, C.c_int64, C.c_int, C.c_int, C.c_int, _P]),
    "ideas_patch_resize": (C.c_int, [_P, _P, C.POINTER(C.c_int)] + [C.c_int] * 8 + [_P]),
    "ideas_patch_resize_bwd": (C.c_int, [_P, _P, C.POINTER(C.c_int)] + [C.c_int] * 9 + [_P]),
    "ideas_sizeof_prep_desc": (C.c_int, []),
    "ideas_weight_prep_batched": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P]),
    "ideas_sizeof_linear_seg": (C.c_int, []),
    "ideas_linear_fwd": (C.c_int, [C.POINTER(LinearSeg), C.c_int, _P, C.c_int, C.c_int, C.c_int, _P]),
    "ideas_linear_bwd_x_workspace": (C.c_int64, [C.c_int, C.c_int, C.c_int]),
    "ideas_linear_bwd_x": (C.c_int, [C.POINTER(LinearSeg), C.c_int, _P, C.c_int, C.c_int, C.c_int, _P, C.c_int64, _P]),
    "ideas_linear_bwd_w": (C.c_int, [C.POINTER(LinearSeg), C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "ideas_act_bwd_dot": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int, C.c_int64, C.c_int, C.c_float, C.c_float, C.c_int, _P]),
}
EXPORTS = tuple(_PROTOS)
ABI_VERSION = 4          # include/ideas_hip.h::IDEAS_ABI_VERSION

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load the shared library (once) and type every entry point.  Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()' "
            "or make -C ideas_amd/csrc).  ideas_amd has no CPU / eager fallback.")
    lib = C.CDLL(LIB_PATH)
    ver = getattr(lib, "ideas_abi_version", None)
    if ver is None or ver() != ABI_VERSION:
        # (checked BEFORE the symbol lookups: a stale library fails with this diagnostic, not with an AttributeError)
        raise RuntimeError(f"libideas_hip.so ABI mismatch: the library reports version {None if ver is None else ver()}, this binding "
                           f"needs {ABI_VERSION} (include/ideas_hip.h); rebuild with make -C ideas_amd/csrc")
    for name, (res, args) in _PROTOS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.ideas_sizeof_conv_params() != C.sizeof(ConvParams):
        raise RuntimeError("libideas_hip.so ABI mismatch (version or ideas_conv_params layout)")
    if lib.ideas_sizeof_prep_desc() != C.sizeof(PrepDesc):
        raise RuntimeError("libideas_hip.so ABI mismatch (ideas_prep_desc layout)")
    if lib.ideas_sizeof_linear_seg() != C.sizeof(LinearSeg):
        raise RuntimeError("libideas_hip.so ABI mismatch (ideas_linear_seg layout)")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().ideas_strerror(rc).decode()
        raise RuntimeError(f"{what} failed: {msg} (code {rc})")


def ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def stream_ptr() -> int:
    # the raw handle of the current stream of the current device, without building a torch.cuda.Stream object (9.5 us a call,
    # 940 calls per iteration: 9 ms of host time of a 190 ms bf16 step -- tools/host_profile.py)
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def make_stream(prio: int) -> "torch.cuda.Stream":
    """A torch view of a HIP stream of the lowest (prio < 0) / default (0) / highest (> 0) priority (ideas_stream_create): torch's own
    pools only offer default and higher."""
    h = C.c_void_p()
    check(load().ideas_stream_create(C.byref(h), int(prio)), "ideas_stream_create")
    return torch.cuda.ExternalStream(h.value)


def require_cuda(*tensors: Optional[torch.Tensor]) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("ideas_amd ops run only on a HIP device tensor (no CPU fallback); got device "
                               f"{t.device}")
