"""ctypes binding of libmmfn_hip.so (the C ABI declared in include/mmfn_hip.h).

The product path fails loudly when the library is missing: there is no CPU or PyTorch fallback.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MMFN_HIP_LIB") or os.path.join(_HERE, "lib", "libmmfn_hip.so")   # MMFN_HIP_LIB: an experimental build

# operand modes / epilogue flags (mirror include/mmfn_hip.h)
A_ROWMAJOR, A_COLMAJOR, A_IM2COL, A_DGRAD = 0, 1, 2, 3
B_NK, B_KN, B_IM2COL, B_DGRADW = 0, 1, 2, 3
EPI_BIAS, EPI_RELU, EPI_GELU, EPI_MASK_AUX, EPI_DROPOUT, EPI_RESIDUAL, EPI_ACCUM, EPI_BF16_OPERANDS, EPI_BF16X3 = 1, 2, 4, 8, 16, 32, 64, 128, 256
EPI_RELU_LAST = 512
EPI_LN_FOLD = 4096
EPI_COLSUM_A = 8192
EPI_KEEP_SLABS = 16384

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
        ("batch", _i32), ("dg_parity", _i32), ("strideA", _i64), ("strideB", _i64), ("strideC", _i64),
        ("ln_c1", _vp), ("ln_mean", _vp), ("ln_rstd", _vp), ("ln_eps", _f32), ("reserved0", _i32),
        ("colsum", _vp),
    ]


class Gemm16Desc(ctypes.Structure):
    """mmfn_gemm16_desc (include/mmfn_hip.h): the bf16-operand GEMM / convolution of the bf16 training mode."""
    _fields_ = [
        ("A", _vp), ("B", _vp), ("C", _vp), ("bias", _vp), ("res", _vp), ("aux", _vp), ("rng_state", _vp), ("workspace", _vp),
        ("stats", _vp), ("bn_y", _vp), ("bn_x", _vp), ("bn_mean", _vp), ("bn_rstd", _vp),
        ("M", _i32), ("N", _i32), ("K", _i32),
        ("lda", _i32), ("ldb", _i32), ("ldc", _i32), ("ldr", _i32), ("ldaux", _i32),
        ("form", _i32),
        ("H", _i32), ("W", _i32), ("Cin", _i32), ("OH", _i32), ("OW", _i32), ("Cout", _i32),
        ("KH", _i32), ("KW", _i32), ("stride", _i32), ("pad", _i32),
        ("flags", _i32), ("splitk", _i32), ("tile", _i32),
        ("rng_stream", ctypes.c_uint32), ("drop_p", _f32), ("stages", _i32), ("stats_mode", _i32), ("reserved", _i32),
    ]


class GptBlockDesc(ctypes.Structure):
    """mmfn_gpt_block_desc (include/mmfn_hip.h): one transformer block of the narrow fusion transformers for the fused kernels."""
    _PTRS = ("ln1_w", "ln1_b", "wqkv", "bqkv", "wproj", "bproj", "ln2_w", "ln2_b", "w1", "b1", "w2", "b2",
             "x", "a", "mu1", "rs1", "qkv", "o", "lse", "x1", "a2", "mu2", "rs2", "h", "x2",
             "g", "gd", "gh", "g1", "gd2", "go", "dqkv", "g_below", "gd_below", "part_ln1", "part_ln2", "rng_state")
    _fields_ = [(n, _vp) for n in _PTRS] + [
        ("B", _i32), ("T", _i32), ("C", _i32), ("NH", _i32),
        ("attn_pdrop", _f32), ("resid_pdrop", _f32), ("eps", _f32),
        ("rng_stream", ctypes.c_uint32), ("rng_stream_below", ctypes.c_uint32), ("below_colsum", _i32), ("reserved", _i32),
    ]


class Conv16HaloDesc(ctypes.Structure):
    """mmfn_conv16_halo_desc (include/mmfn_hip.h): the LDS-resident-patch 3x3 convolution of the bf16 mode."""
    _PTRS = ("x", "w", "out", "stats", "out_res", "bn2_y", "bn2_x", "bn2_mean", "bn2_rstd", "p_mean", "p_rstd", "p_weight", "p_bias",
             "p_means", "p_res", "p_y", "p_x", "a_out", "ge_out")
    _fields_ = [(n, _vp) for n in _PTRS] + [(n, _i32) for n in ("B", "H", "W", "K", "N", "pro", "relu", "flip", "tile", "stages",
                                                                "stats_mode")]


G16_NT, G16_CONV_FWD, G16_CONV_DGRAD, G16_TN, G16_CONV_WGRAD = 0, 1, 2, 3, 4
EPI16_OUT_F32 = 1024
EPI16_RES_F32 = 2048


class MMFNLibraryError(RuntimeError):
    pass


_lib = None

HEADER_PATH = os.path.join(_HERE, "..", "include", "mmfn_hip.h")

_CTYPES = {"int": _i32, "int32_t": _i32, "int64_t": _i64, "float": _f32, "uint32_t": ctypes.c_uint32,
           "void": None}


def _parse_header(path=HEADER_PATH):
    """Derive ctypes signatures from the C declarations so the binding cannot drift from the ABI."""
    import re
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    sigs = {}
    for m in re.finditer(r"\b(int64_t|int)\s+(mmfn_\w+)\s*\(([^)]*)\)\s*;", text):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    argtypes.append(ctypes.POINTER(GemmDesc) if "mmfn_gemm_desc" in a else
                                    (ctypes.POINTER(Gemm16Desc) if "mmfn_gemm16_desc" in a else _vp))
                else:
                    base = a.replace("const", "").split()[0]
                    argtypes.append(_CTYPES[base])
        sigs[name] = (_CTYPES[ret], argtypes)
    return sigs


_SIGNATURES = _parse_header()


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
        if handle.mmfn_sizeof_gemm16_desc() != ctypes.sizeof(Gemm16Desc):
            raise MMFNLibraryError("mmfn_gemm16_desc layout mismatch between C and ctypes")
        if handle.mmfn_sizeof_conv16_halo_desc() != ctypes.sizeof(Conv16HaloDesc):
            raise MMFNLibraryError("mmfn_conv16_halo_desc layout mismatch between C and ctypes")
        if handle.mmfn_sizeof_gpt_block_desc() != ctypes.sizeof(GptBlockDesc):
            raise MMFNLibraryError("mmfn_gpt_block_desc layout mismatch between C and ctypes")
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
