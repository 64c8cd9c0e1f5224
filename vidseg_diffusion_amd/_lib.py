"""ctypes binding of libvidseg_hip.so (the C ABI declared in include/vidseg_hip.h).

The product path has NO CPU fallback: if the library is missing, `lib()` raises.  PyTorch is used
only for device memory and streams; every pointer handed to the library is a raw
`tensor.data_ptr()` and the stream is torch's current HIP stream.
"""
from __future__ import annotations

import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# fp16 activations by default; VIDSEG_ACT=bf16 selects the bfloat16 build of the same sources (see csrc/common.h)
LIB_PATH = os.path.join(_HERE, "libvidseg_hip_bf16.so" if os.environ.get("VIDSEG_ACT", "f16").lower() == "bf16" else "libvidseg_hip.so")
if os.environ.get("VIDSEG_LIB"):                       # kernel experiments (tools/build_exp.py): another build of the same C ABI
    LIB_PATH = os.path.join(_HERE, os.environ["VIDSEG_LIB"])

_P = ctypes.c_void_p
_I = ctypes.c_int
_L = ctypes.c_int64
_U = ctypes.c_uint
_D = ctypes.c_double
_F = ctypes.c_float

# name -> argtypes (return type is always int status, except version/last_error)
_SIGS = {
    "vidseg_mean_normalize_f16": [ctypes.POINTER(_P), _I, _L, _L, _I, _P, _P, _P],
    "vidseg_kmeans_prepare": [_P, _L, _I, _P, _P, _P, _P, _P],
    "vidseg_row_sqnorm_f64": [_P, _L, _I, _P, _P],
    "vidseg_kpp_round_v2": [_P, _P, _P, _L, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P, _P, _P, _P, _P, _L, _P, _P],
    "vidseg_gather_rows_f64": [_P, _P, _I, _P, _I, _P, _P],
    "vidseg_lloyd_iter": [_P, _P, _L, _I, _I, _I, _P, _P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P],
    "vidseg_lloyd_status": [_I, _I, _I, _D, _P, _P, _P, _P, _P],
    "vidseg_lloyd_step": [_P, _P, _P, _L, _I, _I, _I, _I, _D, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "vidseg_kmeans_inertia": [_P, _P, _L, _I, _I, _I, _P, _P, _P, _P, _P],
    "vidseg_add_mean_f64": [_P, _P, _I, _I, _P],
    "vidseg_knn_vote": [_P, _L, _P, _L, _I, _P, _P, _P, _P, _P],
    "vidseg_knn_top4": [_P, _L, _P, _L, _I, _P, _P, _P, _P],
    "vidseg_vote4": [_P, _P, _L, _P, _P],
    "vidseg_track_normalize": [_P, _L, _I, _I, _P, _P],
    "vidseg_track_step": [_P, _I, _I, _I, _I, _I, _I, _P, _I, _P, _P, _P, _P],
    "vidseg_trajectory_vote": [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P],
}

_lib = None


class VidsegError(RuntimeError):
    pass


def register(sigs: dict):
    """Let other modules (UNet operators) add their entry points to the table."""
    _SIGS.update(sigs)
    if _lib is not None:
        _bind(_lib, sigs)


def _bind(l, sigs):
    for name, argtypes in sigs.items():
        fn = getattr(l, name)          # AttributeError if the .so does not export a declared symbol
        fn.argtypes = argtypes
        fn.restype = _I


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VidsegError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the product path.")
        l = ctypes.CDLL(LIB_PATH)
        l.vidseg_version.restype = _I
        l.vidseg_last_error.restype = ctypes.c_char_p
        l.vidseg_act_dtype.restype = _I
        _bind(l, _SIGS)
        _lib = l
    return _lib


def exported_symbols():
    return ["vidseg_version", "vidseg_last_error", "vidseg_act_dtype"] + sorted(_SIGS)


def act_dtype():
    """torch dtype of the UNet path's activations / packed weights as the library was built: float16 (default, the
    reference's CUDA-autocast dtype) or bfloat16 (-DVIDSEG_ACT_BF16)."""
    return torch.float16 if lib().vidseg_act_dtype() == 1 else torch.bfloat16


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    if t is None:
        return None
    return t.data_ptr()


def call(name: str, *args):
    fn = getattr(lib(), name)
    rc = fn(*args)
    if rc != 0:
        raise VidsegError(f"{name} failed ({rc}): {lib().vidseg_last_error().decode()}")


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise VidsegError("vidseg_diffusion_amd operates on HIP device tensors only (no CPU fallback)")
        if t is not None and not t.is_contiguous():
            raise VidsegError("tensor must be contiguous")
