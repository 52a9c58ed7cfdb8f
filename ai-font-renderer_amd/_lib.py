"""ctypes binding of csrc/libafr.so (C ABI declared in include/afr.h).

There is no fallback: if the shared library is missing or does not export every symbol the header
declares, importing this module raises.  `python -c "import __graft_entry__ as g; g.build()"` (or
csrc/build.sh) builds it in-tree for gfx950.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AFR_LIB_PATH") or os.path.join(HERE, "csrc", "libafr.so")   # override: kernel A/B experiments

AFR_KIND_SHEET, AFR_KIND_GLYPH, AFR_KIND_PIXEL = 0, 1, 2
AFR_F32, AFR_BF16 = 0, 1
AFR_TARGET_U8, AFR_TARGET_F32 = 0, 1
AFR_MAX_HIDDEN = 8
BUF_U, BUF_Z, BUF_DZ, BUF_W1T, BUF_W2T, BUF_ACT = 0, 1, 2, 4, 5, 16
GEMM_BIAS, GEMM_RELU, GEMM_RELU_MASK, GEMM_OUT_BF16, GEMM_A_KSTRIDED, GEMM_B_KSTRIDED = 1, 2, 4, 8, 16, 32


class AfrConfig(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("dtype", C.c_int32), ("max_batch", C.c_int32), ("vocab", C.c_int32),
        ("embed_dim", C.c_int32), ("out_h", C.c_int32), ("out_w", C.c_int32),
        ("max_length", C.c_int32), ("heads", C.c_int32), ("fc_dim", C.c_int32),
        ("p_embed", C.c_float), ("p_attn", C.c_float), ("p_fc", C.c_float), ("ln_eps", C.c_float),
        ("n_hidden", C.c_int32), ("hidden", C.c_int32 * AFR_MAX_HIDDEN), ("n_fonts", C.c_int32),
        ("seed", C.c_uint64), ("rank", C.c_int32), ("reserved", C.c_int32),
    ]


_vp, _i32, _i64, _f32, _u64, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_uint64, C.c_size_t

# name -> (restype, argtypes): every function include/afr.h declares
SIGNATURES = {
    "afr_version": (_i32, []),
    "afr_last_error": (C.c_char_p, []),
    "afr_plan_create": (_i32, [C.POINTER(AfrConfig), C.POINTER(_vp)]),
    "afr_plan_destroy": (_i32, [_vp]),
    "afr_param_elems": (_i64, [_vp]),
    "afr_param_count": (_i32, [_vp]),
    "afr_param_info": (_i32, [_vp, _i32, C.c_char_p, _i32, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(C.c_int32), C.POINTER(_i64)]),
    "afr_workspace_bytes": (_sz, [_vp]),
    "afr_bind": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _sz]),
    "afr_sync_params": (_i32, [_vp, _vp]),
    "afr_forward": (_i32, [_vp, _vp, _vp, _i32, _i32, _vp, _i32, _u64, _vp]),
    "afr_loss_grad": (_i32, [_vp, _vp, _i32, _i32, _i64, _vp, _vp]),
    "afr_set_output_grad": (_i32, [_vp, _vp, _i32, _vp]),
    "afr_backward": (_i32, [_vp, _vp]),
    "afr_backward_stages": (_i32, [_vp]),
    "afr_backward_stage": (_i32, [_vp, _i32, C.POINTER(_i64), C.POINTER(_i64), _vp]),
    "afr_forward_loss": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i64, _vp, _u64, _vp]),
    "afr_adamw_step": (_i32, [_vp, _f32, _f32, _f32, _f32, _f32, _i64, _f32, _vp]),
    "afr_train_step": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i64, _vp, _u64, _i32, _f32, _f32, _f32, _f32, _f32, _i64, _vp]),
    "afr_error_flags": (_i32, [_vp, _vp, C.POINTER(C.c_uint32)]),
    "afr_profile_dominant": (_i32, [_vp, _i32]),
    "afr_profile_read": (_i32, [_vp, C.c_char_p, _i32, C.POINTER(C.c_double), C.POINTER(_i64), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "afr_profile_dump": (_i32, [_vp, C.c_char_p, _i32]),
    "afr_debug_copy": (_i32, [_vp, _i32, _vp, _sz, C.POINTER(_sz), _vp]),
    "afr_debug_sheet_gather": (_i32, [_vp, _vp, _i32, _i32, _vp, _vp]),
    "afr_op_gemm": (_i32, [_i32, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "afr_op_gemm_fix_workspace_bytes": (_sz, [_i32, _i32, _i32, _i32]),
    "afr_op_gemm_fix": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _sz, _vp]),
    "afr_op_gemm_pair_plan": (_i32, [_i32, _i32, _i32, C.POINTER(_i32), C.POINTER(_sz)]),
    "afr_op_gemm_pair": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _sz, _vp]),
    "afr_op_reduce": (_i32, [_vp, _vp, _i32, _i64, _i64, _f32, _i32, _vp]),
    "afr_op_reduce_group": (_i32, [_i32, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_i32), C.POINTER(_i64), C.POINTER(_i64), _vp]),
    "afr_op_adamw": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _i64, _f32, _vp]),
    "afr_op_mse_grad": (_i32, [_i32, _vp, _vp, _i32, _vp, _i64, _i64, _i64, _vp, _vp, _vp]),
    "afr_op_f32_to_bf16": (_i32, [_vp, _vp, _i64, _vp]),
    "afr_op_f32_to_fp8": (_i32, [_vp, _vp, _i64, _f32, _vp]),
    "afr_op_gemm_fp8": (_i32, [_i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _vp]),
}


class AfrError(RuntimeError):
    pass


def load(path=LIB_PATH):
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: the HIP extension is the only implementation of the hot path "
            "(no CPU fallback).  Build it with ai-font-renderer_amd/csrc/build.sh or __graft_entry__.build().")
    # torch first: it ships its own libamdhip64, and libafr.so must bind to THAT runtime instance (the device pointers
    # it is handed come from torch's allocator).  Loaded before torch, libafr.so would pull in /opt/rocm's copy and the
    # process would hold two HIP runtimes ("no ROCm-capable device is detected" at the first launch).
    import torch  # noqa: F401
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = load()
    return _lib


def check(rc):
    if rc != 0:
        raise AfrError(f"libafr error {rc}: {lib().afr_last_error().decode()}")
