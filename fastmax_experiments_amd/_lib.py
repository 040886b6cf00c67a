"""ctypes binding of the C ABI declared in include/fastmax_hip.h.

Fails loudly: if libfastmax_hip.so is absent or a symbol is missing, importing the operator
raises -- there is no eager/PyTorch/CPU fallback for the hot path.
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# FASTMAX_LIB_PATH: A/B runs of tools/ against another build of the same library (never a different implementation)
LIB_PATH = os.environ.get("FASTMAX_LIB_PATH") or os.path.join(HERE, "libfastmax_hip.so")

F32, BF16, F16 = 0, 1, 2
PATH_AUTO, PATH_QUADRATIC, PATH_RECURRENT, PATH_MFMA, PATH_QUADRATIC_MFMA = 0, 1, 2, 3, 4
PATH_NAMES = {PATH_AUTO: "auto", PATH_QUADRATIC: "quadratic", PATH_RECURRENT: "recurrent", PATH_MFMA: "mfma", PATH_QUADRATIC_MFMA: "quadratic_mfma"}
E_BAD_P = -1
E_BAD_SHAPE = -2

# every symbol include/fastmax_hip.h declares
SYMBOLS = ["fastmax_hip_forward_workspace", "fastmax_hip_forward", "fastmax_hip_backward_workspace",
           "fastmax_hip_backward", "fastmax_hip_normalize_workspace", "fastmax_hip_normalize",
           "fastmax_hip_abi_version", "fastmax_hip_select_path", "fastmax_hip_error_string",
           "fastmax_hip_decode_state_bytes", "fastmax_hip_p1_prefill_state", "fastmax_hip_p1_decode_step",
           "fastmax_hip_normalize_stats", "fastmax_hip_normalize_cast", "fastmax_hip_normalize_backward_workspace",
           "fastmax_hip_normalize_backward", "fastmax_hip_rope_qkv_split", "fastmax_hip_rope_qkv_split_backward", "fastmax_hip_cross_entropy_forward", "fastmax_hip_cross_entropy_backward",
           "fastmax_hip_linearmax_forward", "fastmax_hip_linearmax_forward_auto", "fastmax_hip_linearmax_forward_auto_workspace",
           "fastmax_hip_linearmax_backward", "fastmax_hip_linearmax_train_supported",
           "fastmax_hip_nf4_linear_forward", "fastmax_hip_nf4_linear_backward_input", "fastmax_hip_nf4_dequantize",
           "fastmax_hip_lora_down", "fastmax_hip_lora_tn_workspace", "fastmax_hip_lora_tn", "fastmax_hip_lora_up",
           "fastmax_hip_forward_state_bytes", "fastmax_hip_backward_with_states",
           "fastmax_hip_lora_scatter", "fastmax_hip_lora_scatter_backward",
           "fastmax_hip_normalize_cast_expand", "fastmax_hip_normalize_backward_expand", "fastmax_hip_tune",
           "fastmax_hip_nf4_linear_forward_s", "fastmax_hip_nf4_linear_backward_input_s", "fastmax_hip_nf4_dequantize_s",
           "fastmax_hip_qlora_gemm", "fastmax_hip_nf4_dequantize_transposed",
           "fastmax_hip_qlora_gemm_rope", "fastmax_hip_tune_get", "fastmax_hip_build_flags",
           "fastmax_hip_normalize_stats2_workspace", "fastmax_hip_normalize_stats2",
           "fastmax_hip_lora_down_dropout", "fastmax_hip_lora_tn_dropout", "fastmax_hip_lora_up_dropout", "fastmax_hip_lora_dropout_mask"]


class Problem(ctypes.Structure):
    _fields_ = [("B", ctypes.c_int), ("H", ctypes.c_int), ("Nq", ctypes.c_int), ("Nk", ctypes.c_int),
                ("D", ctypes.c_int), ("in_dtype", ctypes.c_int), ("out_dtype", ctypes.c_int), ("p", ctypes.c_int),
                ("causal", ctypes.c_int), ("a", ctypes.c_float), ("b", ctypes.c_float), ("g0", ctypes.c_float),
                ("path", ctypes.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m fastmax_experiments_amd.build` "
            "(hipcc, --offload-arch=gfx950). The fastmax operator has no fallback path.")
    L = ctypes.CDLL(LIB_PATH)
    for s in SYMBOLS:
        if not hasattr(L, s):
            raise RuntimeError(f"libfastmax_hip.so does not export {s}")
    vp, i64p, fp, sz, ci = ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64), ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
    pp = ctypes.POINTER(Problem)
    L.fastmax_hip_tune.argtypes = [ctypes.c_char_p, ci]
    L.fastmax_hip_tune.restype = ci
    L.fastmax_hip_tune_get.argtypes = [ctypes.c_char_p]
    L.fastmax_hip_tune_get.restype = ci
    L.fastmax_hip_build_flags.argtypes = []
    L.fastmax_hip_build_flags.restype = ci
    L.fastmax_hip_forward_workspace.argtypes = [pp]
    L.fastmax_hip_forward_workspace.restype = sz
    L.fastmax_hip_forward.argtypes = [pp, vp, i64p, vp, i64p, vp, i64p, vp, fp, vp, sz, vp]
    L.fastmax_hip_forward.restype = ci
    L.fastmax_hip_backward_workspace.argtypes = [pp]
    L.fastmax_hip_backward_workspace.restype = sz
    L.fastmax_hip_backward.argtypes = [pp, vp, i64p, vp, i64p, vp, i64p, vp, fp, vp, i64p, vp, vp, vp, vp, sz, vp]
    L.fastmax_hip_backward_with_states.argtypes = [pp, vp, i64p, vp, i64p, vp, i64p, vp, fp, vp, i64p, vp, vp, vp, vp, sz, vp, sz, vp]
    L.fastmax_hip_backward_with_states.restype = ci
    L.fastmax_hip_forward_state_bytes.argtypes = [pp, vp, i64p, vp, i64p, vp, i64p, vp]
    L.fastmax_hip_forward_state_bytes.restype = sz
    L.fastmax_hip_backward.restype = ci
    L.fastmax_hip_normalize_workspace.argtypes = [ci, ci]
    L.fastmax_hip_normalize_workspace.restype = sz
    L.fastmax_hip_normalize.argtypes = [vp, i64p, ci, fp, fp, ci, ci, ci, ci, vp, sz, vp]
    L.fastmax_hip_normalize.restype = ci
    L.fastmax_hip_normalize_stats.argtypes = [vp, i64p, ci, fp, ci, ci, ci, ci, vp, sz, vp]
    L.fastmax_hip_normalize_stats.restype = ci
    L.fastmax_hip_normalize_stats2_workspace.argtypes = [ci, ci, ci]
    L.fastmax_hip_normalize_stats2_workspace.restype = sz
    L.fastmax_hip_normalize_stats2.argtypes = [vp, i64p, vp, i64p, ci, fp, fp, ci, ci, ci, ci, vp, sz, vp]
    L.fastmax_hip_normalize_stats2.restype = ci
    L.fastmax_hip_normalize_cast.argtypes = [vp, i64p, ci, vp, fp, ci, ci, ci, ci, vp, sz, vp]
    L.fastmax_hip_normalize_cast.restype = ci
    L.fastmax_hip_normalize_backward_workspace.argtypes = [ci, ci, ci]
    L.fastmax_hip_normalize_backward_workspace.restype = sz
    L.fastmax_hip_normalize_backward.argtypes = [vp, i64p, ci, vp, fp, vp, ci, ci, ci, ci, vp, sz, vp]
    L.fastmax_hip_normalize_backward.restype = ci
    L.fastmax_hip_rope_qkv_split.argtypes = [vp, fp, fp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, vp]
    L.fastmax_hip_rope_qkv_split.restype = ci
    L.fastmax_hip_rope_qkv_split_backward.argtypes = [vp, vp, vp, fp, fp, vp, ci, ci, ci, ci, ci, ci, ci, ci, vp]
    L.fastmax_hip_rope_qkv_split_backward.restype = ci
    i64 = ctypes.c_int64
    L.fastmax_hip_cross_entropy_forward.argtypes = [vp, i64, vp, fp, fp, i64, ci, i64, ci, vp]
    L.fastmax_hip_cross_entropy_forward.restype = ci
    L.fastmax_hip_cross_entropy_backward.argtypes = [vp, i64, vp, fp, fp, ctypes.c_float, vp, i64, i64, ci, i64, ci, vp]
    L.fastmax_hip_cross_entropy_backward.restype = ci
    L.fastmax_hip_linearmax_forward.argtypes = [pp, vp, i64p, vp, i64p, vp, i64p, fp, fp, vp, fp, vp, sz, vp]
    L.fastmax_hip_linearmax_forward.restype = ci
    L.fastmax_hip_linearmax_forward_auto.argtypes = [pp, vp, i64p, vp, i64p, vp, i64p, fp, fp, vp, vp, vp, fp, vp, sz, vp]
    L.fastmax_hip_linearmax_forward_auto.restype = ci
    L.fastmax_hip_linearmax_forward_auto_workspace.argtypes = [pp]
    L.fastmax_hip_linearmax_forward_auto_workspace.restype = sz
    L.fastmax_hip_linearmax_train_supported.argtypes = [pp]
    L.fastmax_hip_linearmax_train_supported.restype = ci
    L.fastmax_hip_linearmax_backward.argtypes = [pp, vp, i64p, vp, i64p, vp, i64p, vp, fp, vp, i64p, fp, fp, vp, vp, vp, vp, vp, vp, sz, vp, sz, ci, vp]
    L.fastmax_hip_linearmax_backward.restype = ci
    i64 = ctypes.c_int64
    L.fastmax_hip_decode_state_bytes.argtypes = [ci, ci, ci]
    L.fastmax_hip_decode_state_bytes.restype = sz
    L.fastmax_hip_p1_prefill_state.argtypes = [pp, vp, i64p, vp, i64p, fp, vp]
    L.fastmax_hip_p1_prefill_state.restype = ci
    L.fastmax_hip_p1_decode_step.argtypes = [vp, i64p, vp, i64p, vp, i64p, fp, vp, ci, ci, ci, ci, ci, ctypes.c_float, i64, vp]
    L.fastmax_hip_p1_decode_step.restype = ci
    L.fastmax_hip_nf4_linear_forward.argtypes = [vp, i64, vp, vp, vp, vp, vp, vp, i64, ci, ci, ci, ci, vp]
    L.fastmax_hip_nf4_linear_forward.restype = ci
    L.fastmax_hip_nf4_linear_backward_input.argtypes = [vp, i64, vp, vp, vp, i64, ci, ci, ci, ci, vp]
    L.fastmax_hip_nf4_linear_backward_input.restype = ci
    L.fastmax_hip_nf4_dequantize.argtypes = [vp, vp, vp, i64, ci, vp]
    L.fastmax_hip_nf4_dequantize.restype = ci
    L.fastmax_hip_nf4_linear_forward_s.argtypes = [vp, i64, vp, vp, vp, vp, vp, vp, i64, ci, ci, ci, ci, vp]
    L.fastmax_hip_nf4_linear_forward_s.restype = ci
    L.fastmax_hip_nf4_linear_backward_input_s.argtypes = [vp, i64, vp, vp, vp, i64, ci, ci, ci, ci, vp]
    L.fastmax_hip_nf4_linear_backward_input_s.restype = ci
    L.fastmax_hip_nf4_dequantize_s.argtypes = [vp, vp, vp, i64, ci, vp]
    L.fastmax_hip_nf4_dequantize_s.restype = ci
    L.fastmax_hip_qlora_gemm.argtypes = [vp, i64, vp, ci, vp, vp, vp, vp, ci, vp, i64, ci, ci, ci, vp]
    L.fastmax_hip_qlora_gemm.restype = ci
    L.fastmax_hip_nf4_dequantize_transposed.argtypes = [vp, vp, vp, ci, ci, vp]
    L.fastmax_hip_nf4_dequantize_transposed.restype = ci
    L.fastmax_hip_qlora_gemm_rope.argtypes = [vp, i64, vp, vp, vp, vp, ci, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, ci, vp]
    L.fastmax_hip_qlora_gemm_rope.restype = ci
    L.fastmax_hip_lora_down.argtypes = [vp, i64, vp, i64, vp, i64, vp, i64, ci, ci, ci, vp]
    L.fastmax_hip_lora_down.restype = ci
    L.fastmax_hip_lora_tn_workspace.argtypes = [ci, ci, ci]
    L.fastmax_hip_lora_tn_workspace.restype = i64
    L.fastmax_hip_lora_tn.argtypes = [vp, i64, vp, i64, vp, ci, ci, ci, vp, ci, ci, ci, vp]
    L.fastmax_hip_lora_tn.restype = ci
    L.fastmax_hip_lora_up.argtypes = [vp, i64, vp, i64, vp, i64, ci, vp, ci, ci, ci, vp]
    cf = ctypes.c_float
    L.fastmax_hip_lora_down_dropout.argtypes = [vp, i64, vp, i64, vp, i64, vp, i64, ci, ci, ci, vp, cf, vp]
    L.fastmax_hip_lora_down_dropout.restype = ci
    L.fastmax_hip_lora_tn_dropout.argtypes = [vp, i64, vp, i64, vp, ci, ci, ci, vp, ci, ci, ci, vp, cf, vp]
    L.fastmax_hip_lora_tn_dropout.restype = ci
    L.fastmax_hip_lora_up_dropout.argtypes = [vp, i64, vp, i64, vp, i64, ci, vp, ci, ci, ci, vp, cf, vp]
    L.fastmax_hip_lora_up_dropout.restype = ci
    L.fastmax_hip_lora_dropout_mask.argtypes = [vp, ci, ci, vp, cf, vp]
    L.fastmax_hip_lora_dropout_mask.restype = ci
    L.fastmax_hip_normalize_cast_expand.argtypes = [vp, i64p, ci, vp, vp, ci, ci, ci, ci, ci, vp, sz, vp]
    L.fastmax_hip_normalize_cast_expand.restype = ci
    L.fastmax_hip_normalize_backward_expand.argtypes = [vp, i64p, ci, vp, vp, vp, ci, ci, ci, ci, ci, vp, sz, vp]
    L.fastmax_hip_normalize_backward_expand.restype = ci
    L.fastmax_hip_lora_scatter.argtypes = [vp, ci, ci, vp, ci, ctypes.c_float, vp, i64, ci, ci, vp]
    L.fastmax_hip_lora_scatter.restype = ci
    L.fastmax_hip_lora_scatter_backward.argtypes = [vp, ci, i64, vp, vp, ctypes.c_float, vp, ci, ci, ci, vp]
    L.fastmax_hip_lora_scatter_backward.restype = ci
    L.fastmax_hip_lora_up.restype = ci
    L.fastmax_hip_abi_version.restype = ci
    L.fastmax_hip_select_path.argtypes = [pp]
    L.fastmax_hip_select_path.restype = ci
    L.fastmax_hip_error_string.argtypes = [ci]
    L.fastmax_hip_error_string.restype = ctypes.c_char_p
    if L.fastmax_hip_abi_version() != 8:
        raise RuntimeError("libfastmax_hip.so ABI version mismatch")
    _lib = L
    return L


def error_string(code):
    return lib().fastmax_hip_error_string(int(code)).decode()


def check(code, what):
    if code == 0:
        return
    if code == E_BAD_P:
        raise ValueError(f"{what}: {error_string(code)}")
    raise RuntimeError(f"{what} failed: rc={code} ({error_string(code)})")
