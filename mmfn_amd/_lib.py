"""ctypes binding of libmmfn_hip.so (the C ABI declared in include/mmfn_hip.h).

The product path fails loudly when the library is missing: there is no CPU or PyTorch fallback.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libmmfn_hip.so")

# operand modes / epilogue flags (mirror include/mmfn_hip.h)
A_ROWMAJOR, A_COLMAJOR, A_IM2COL, A_DGRAD = 0, 1, 2, 3
B_NK, B_KN, B_IM2COL, B_DGRADW = 0, 1, 2, 3
EPI_BIAS, EPI_RELU, EPI_GELU, EPI_MASK_AUX, EPI_DROPOUT, EPI_RESIDUAL, EPI_ACCUM = 1, 2, 4, 8, 16, 32, 64

_vp = ctypes.c_void_p
_i32 = ctypes.c_int32
_i64 = ctypes.c_int64
_f32 = ctypes.c_float


class GemmDesc(ctypes.Structure):
    _fields_ = [
        ("A", _vp), ("B", _vp), ("C", _vp), ("bias", _vp), ("res", _vp), ("aux", _vp),
        ("rng_state", _vp), ("workspace", _vp),
        ("M", _i32), ("N", _i32), ("K", _i32),
        ("lda", _i32), ("ldb", _i32), ("ldc", _i32), ("ldr", _i32), ("ldaux", _i32),
        ("a_mode", _i32), ("b_mode", _i32),
        ("H", _i32), ("W", _i32), ("Cin", _i32), ("OH", _i32), ("OW", _i32), ("Cout", _i32),
        ("KH", _i32), ("KW", _i32), ("stride", _i32), ("pad", _i32),
        ("flags", _i32), ("splitk", _i32), ("tile", _i32),
        ("rng_stream", ctypes.c_uint32), ("drop_p", _f32),
    ]


class MMFNLibraryError(RuntimeError):
    pass


_lib = None

# name -> (restype, argtypes); every symbol declared in include/mmfn_hip.h is listed here and
# tests/test_abi.py checks the two stay in sync.
_SIGNATURES = {
    "mmfn_abi_version": (_i32, []),
    "mmfn_sizeof_gemm_desc": (_i32, []),
    "mmfn_device_selftest": (_i32, [_vp]),
    "mmfn_fill_f32": (_i32, [_vp, _f32, _i64, _vp]),
    "mmfn_axpby_f32": (_i32, [_vp, _vp, _f32, _f32, _i64, _vp]),
    "mmfn_rng_advance": (_i32, [_vp, _vp]),
    "mmfn_gemm_f32": (_i32, [ctypes.POINTER(GemmDesc), _vp]),
    "mmfn_gemm_workspace_bytes": (_i64, [ctypes.POINTER(GemmDesc)]),
}


def lib():
    """Load (once) and return the ctypes handle; raises if the HIP library is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MMFNLibraryError(
                "libmmfn_hip.so not found at %s — run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback for the MMFN HIP path)" % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        if handle.mmfn_sizeof_gemm_desc() != ctypes.sizeof(GemmDesc):
            raise MMFNLibraryError("mmfn_gemm_desc layout mismatch between C and ctypes")
        _lib = handle
    return _lib


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def check(code, what):
    if code != 0:
        raise MMFNLibraryError("%s failed with code %d" % (what, code))
