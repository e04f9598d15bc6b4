"""ctypes binding of liblemevit_hip.so (the C ABI declared in include/lemevit_hip.h).

There is deliberately NO fallback: if the shared library is missing or does not load, importing
this module raises, and every op in ``lemevit_amd.ops`` is unusable.  Build it with
``python -c 'import __graft_entry__ as g; g.build()'`` or ``make -C lemevit_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  -- FIRST: PyTorch-ROCm ships its own HIP runtime; the library must bind to the copy torch has already loaded (loaded before torch -- `import lemevit_amd`
                            # in a fresh interpreter -- it pulled /opt/rocm's runtime and every launch later failed with "no ROCm-capable device is detected")

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LMV_LIB_PATH") or os.path.join(_HERE, "csrc", "liblemevit_hip.so")          # LMV_LIB_PATH: another build of the same library (A/B runs of kernel variants inside one gpurun call)

LMV_F32, LMV_BF16 = 0, 1
ACT_NONE, ACT_GELU, ACT_GELU_GRAD = 0, 1, 2
ABI_VERSION = 12


class LinearProblem(C.Structure):
    _fields_ = [("a", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p), ("res", C.c_void_p),
                ("row_scale", C.c_void_p), ("aux", C.c_void_p), ("out", C.c_void_p), ("out_pre", C.c_void_p),
                ("bias_grad", C.c_void_p), ("rows", C.c_int64), ("rows_per_sample", C.c_int32), ("_pad", C.c_int32)]


class MlpProblem(C.Structure):
    _fields_ = [("x", C.c_void_p), ("out", C.c_void_p), ("row_scale", C.c_void_p), ("rows", C.c_int64), ("rows_per_sample", C.c_int32), ("_pad", C.c_int32)]


class MlpWeights(C.Structure):
    _fields_ = [("w1f", C.c_void_p), ("colsum1", C.c_void_p), ("b1f", C.c_void_p), ("w2", C.c_void_p), ("b2", C.c_void_p)]


class ReduceSeg(C.Structure):
    _fields_ = [("ws", C.c_void_p), ("out_w", C.c_void_p), ("out_b", C.c_void_p), ("slab_stride", C.c_int64), ("nw", C.c_int64),
                ("nslabs", C.c_int32), ("nb", C.c_int32), ("kind", C.c_int32), ("mode", C.c_int32)]


class LnSegment(C.Structure):
    _fields_ = [("x", C.c_void_p), ("y", C.c_void_p), ("stats", C.c_void_p), ("dy", C.c_void_p), ("dres", C.c_void_p),
                ("dx", C.c_void_p), ("rows", C.c_int64), ("dx_scale", C.c_void_p), ("dx_scaled", C.c_void_p), ("rows_per_sample", C.c_int64)]


class RowScaleSegment(C.Structure):
    _fields_ = [("x", C.c_void_p), ("scale", C.c_void_p), ("y", C.c_void_p), ("rows", C.c_int64), ("rows_per_sample", C.c_int)]


class AttnDesc(C.Structure):
    _fields_ = [("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p), ("o", C.c_void_p), ("lse", C.c_void_p),
                ("d_o", C.c_void_p), ("dq", C.c_void_p), ("dk", C.c_void_p), ("dv", C.c_void_p),
                ("q_bs", C.c_int64), ("q_rs", C.c_int64), ("k_bs", C.c_int64), ("k_rs", C.c_int64),
                ("v_bs", C.c_int64), ("v_rs", C.c_int64), ("o_bs", C.c_int64), ("o_rs", C.c_int64),
                ("B", C.c_int32), ("H", C.c_int32), ("Lq", C.c_int32), ("Lk", C.c_int32),
                ("scale", C.c_float), ("_pad", C.c_int32)]


class BlockDesc(C.Structure):
    _fields_ = ([(n, C.c_int32) for n in ("kind", "dtype", "B", "H", "W", "M", "C", "hidden")] + [("eps", C.c_float), ("flags", C.c_int32)] +
                [(n, C.c_void_p) for n in ("pos_w", "pos_b", "n1_w", "n1_b")] + [("attn_w", C.c_void_p * 4), ("attn_b", C.c_void_p * 4)] +
                [(n, C.c_void_p) for n in ("n2_w", "n2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b")] + [("masks", C.c_void_p * 4)] +
                [(n, C.c_void_p) for n in ("g_pos_w", "g_pos_b", "g_n1_w", "g_n1_b")] + [("g_attn_w", C.c_void_p * 4), ("g_attn_b", C.c_void_p * 4)] +
                [(n, C.c_void_p) for n in ("g_n2_w", "g_n2_b", "g_fc1_w", "g_fc1_b", "g_fc2_w", "g_fc2_b")] +
                [("fold_attn_w", C.c_void_p * 2), ("fold_attn_s", C.c_void_p * 2), ("fold_attn_b", C.c_void_p * 2)] +
                [(n, C.c_void_p) for n in ("fold_fc1_w", "fold_fc1_s", "fold_fc1_b", "fc2_wt", "fc1_wt")] + [("attn_wt", C.c_void_p * 2)])


class SStageBlockParams(C.Structure):
    _fields_ = ([(n, C.c_int32) for n in ("C", "heads", "hidden", "_pad")] +
                [(n, C.c_void_p) for n in ("qkv_w", "proj_w", "fc1_w", "fc2_w", "n1_w", "n1_b", "qkv_b", "proj_b", "n2_w", "n2_b", "fc1_b", "fc2_b", "pos_w", "pos_b")])


class SStageDesc(C.Structure):
    _fields_ = ([(n, C.c_int32) for n in ("B", "H", "W", "M", "C", "heads", "hidden", "nblocks", "dtype")] + [("eps", C.c_float), ("wpk", C.c_void_p), ("vec", C.c_void_p), ("timing", C.c_void_p), ("timing_block", C.c_int32), ("kind", C.c_int32)])


class DStageBlockParams(C.Structure):
    _fields_ = ([(n, C.c_int32) for n in ("C", "heads", "hidden", "_pad")] +
                [(n, C.c_void_p) for n in ("qkv1_w", "qkv2_w", "projx_w", "projc_w", "fc1_w", "fc2_w", "n1_w", "n1_b", "qkv1_b", "qkv2_b", "projx_b", "projc_b",
                                           "n2_w", "n2_b", "fc1_b", "fc2_b", "pos_w", "pos_b")])


class TransposeSeg(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("rows", C.c_int32), ("cols", C.c_int32)]


_P, _I, _L, _F, _Z = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t

# name -> (restype, argtypes); the complete export list of include/lemevit_hip.h
SIGNATURES = {
    "lmv_abi_version": (_I, []),
    "lmv_config_set": (_I, [C.c_char_p, _I]),
    "lmv_config_get": (_I, [C.c_char_p, C.POINTER(C.c_int)]),
    "lmv_last_error": (C.c_char_p, []),
    "lmv_linear_fwd": (_I, [C.POINTER(LinearProblem), _I, _I, _I, _I, _I, _P]),
    "lmv_linear_dx": (_I, [C.POINTER(LinearProblem), _I, _I, _I, _I, _I, _P]),
    "lmv_linear_dw_workspace_bytes": (_Z, [C.POINTER(LinearProblem), _I, _I, _I, _I]),
    "lmv_linear_dw": (_I, [C.POINTER(LinearProblem), _I, _I, _I, _P, _Z, _I, _P]),
    "lmv_linear_dw_partial": (_I, [C.POINTER(LinearProblem), _I, _I, _I, _P, _Z, _I, _P, C.POINTER(ReduceSeg), C.POINTER(C.c_int)]),
    "lmv_reduce_batch": (_I, [C.POINTER(ReduceSeg), _I, _P]),
    "lmv_ln_fold": (_I, [_P, _P, _P, _P, _I, _I, _P, _P, _P, _I, _P]),
    "lmv_ln_linear_fwd": (_I, [C.POINTER(LinearProblem), _I, _I, _I, _F, _I, _I, _P]),
    "lmv_mlp_fused_supported": (_I, [_I, _I, _I]),
    "lmv_gelu_poly_eval": (_I, [_P, _P, C.c_int64, _P]),
    "lmv_mlp_fused_fwd": (_I, [C.POINTER(MlpProblem), _I, C.POINTER(MlpWeights), _I, _I, _F, _I, _P]),
    "lmv_attn_out_proj_residual": (_I, [C.POINTER(LinearProblem), _I, _I, _I, _P]),
    "lmv_layernorm_fwd": (_I, [C.POINTER(LnSegment), _I, _P, _P, _I, _F, _I, _P]),
    "lmv_layernorm_gelu_fwd": (_I, [C.POINTER(LnSegment), _I, _P, _P, _I, _F, _I, _P]),
    "lmv_layernorm_gelu_bwd": (_I, [C.POINTER(LnSegment), _I, _P, _P, _P, _P, _I, _P, _Z, _I, _P]),
    "lmv_layernorm_bwd_workspace_bytes": (_Z, [_L, _I, _I]),
    "lmv_layernorm_bwd": (_I, [C.POINTER(LnSegment), _I, _P, _P, _P, _I, _P, _Z, _I, _P]),
    "lmv_layernorm_bwd_partial": (_I, [C.POINTER(LnSegment), _I, _P, _I, _P, _Z, C.POINTER(C.c_int), _I, _P]),
    "lmv_layernorm_bwd_reduce": (_I, [_P, _I, _I, _P, _P, _P]),
    "lmv_ln_linear_exact_fwd_supported": (_I, [_I, _I, _I]),
    "lmv_ln_linear_exact_fwd": (_I, [C.POINTER(LinearProblem), C.POINTER(LnSegment), _I, _I, _I, _P, _P, _F, _I, _P]),
    "lmv_linear_res_ln_fwd_supported": (_I, [_I, _I, _I]),
    "lmv_linear_res_ln_fwd": (_I, [C.POINTER(LinearProblem), C.POINTER(LnSegment), _I, _I, _I, _P, _P, _F, _I, _P]),
    "lmv_linear_dx_ln_bwd_supported": (_I, [_I, _I, _I]),
    "lmv_linear_dx_ln_bwd_workspace_bytes": (_Z, [_L, _I]),
    "lmv_linear_dx_ln_bwd": (_I, [C.POINTER(LinearProblem), C.POINTER(LnSegment), _I, _I, _I, _P, _P, _Z, C.POINTER(C.c_int), _I, _P]),
    "lmv_batchnorm_workspace_bytes": (_Z, [_I]),
    "lmv_batchnorm_train_fwd": (_I, [_P, _P, _P, _P, _P, _F, _F, _I, _P, _P, _L, _I, _P, _Z, _I, _P]),
    "lmv_batchnorm_train_bwd": (_I, [_P, _P, _P, _P, _P, _I, _P, _P, _P, _L, _I, _P, _Z, _I, _P]),
    "lmv_dwconv3x3_residual_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "lmv_dwconv3x3_residual_bwd_data": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "lmv_dwconv3x3_bwd_weight_workspace_bytes": (_Z, [_I, _I, _I, _I, _I]),
    "lmv_dwconv3x3_bwd_weight": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P, _Z, _I, _P]),
    "lmv_dwconv3x3_bwd_weight_partial": (_I, [_P, _P, _I, _I, _I, _I, _P, _Z, C.POINTER(C.c_int), _I, _P]),
    "lmv_attn_workspace_bytes": (_Z, [_I, _I, _I, _I, _I]),
    "lmv_attn_fwd": (_I, [C.POINTER(AttnDesc), _P, _Z, _I, _P]),
    "lmv_attn_bwd": (_I, [C.POINTER(AttnDesc), _P, _Z, _I, _P]),
    "lmv_attn_fwd_pair": (_I, [C.POINTER(AttnDesc), _P, _Z, _I, _P]),
    "lmv_attn_bwd_pair": (_I, [C.POINTER(AttnDesc), _P, _Z, _I, _P]),
    "lmv_sa_core_fwd": (_I, [_P, _P, _P, _I, _I, _I, _P, _Z, _I, _P]),
    "lmv_ca_core_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P, _Z, _I, _P]),
    "lmv_dca_core_fwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _Z, _I, _P]),
    "lmv_cast": (_I, [_P, _I, _P, _I, _L, _P]),
    "lmv_im2col3x3s2_c3": (_I, [_P, _I, _P, _I, _I, _I, _I, _L, _L, _L, _L, _P]),
    "lmv_row_scale_multi": (_I, [_P, _I, _I, _I, _P]),
    "lmv_transpose_batch": (_I, [C.POINTER(TransposeSeg), _I, _I, _P]),
    "lmv_row_scale": (_I, [_P, _P, _P, _L, _I, _I, _I, _P]),
    "lmv_im2col3x3s2_nhwc": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "lmv_conv3x3s2_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "lmv_conv3x3s2_dw_workspace_bytes": (_Z, [_I, _I, _I, _I, _I, _I, _I]),
    "lmv_conv3x3s2_dw": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _Z, _I, _P]),
    "lmv_col2im3x3s2_nhwc": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "lmv_token_mean2_fwd": (_I, [_P, _I, _P, _I, _I, _I, _P, _I, _P]),
    "lmv_token_mean2_affine_fwd": (_I, [_P, _I, _P, _I, _I, _I, _P, _P, _P, _I, _P]),
    "lmv_token_mean2_bwd": (_I, [_P, _P, _I, _P, _I, _I, _I, _I, _P]),
    "lmv_adamw_flat": (_I, [_P, _P, _P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _I, _P, _P]),
    "lmv_ema_flat": (_I, [_P, _P, _L, _F, _P]),
    "lmv_block_arena_bytes": (_Z, [C.POINTER(BlockDesc)]),
    "lmv_block_bwd_scratch_bytes": (_Z, [C.POINTER(BlockDesc)]),
    "lmv_block_fwd": (_I, [C.POINTER(BlockDesc), _P, _P, _P, _P, _P, _Z, _I, _P]),
    "lmv_block_fwd_range": (_I, [C.POINTER(BlockDesc), _P, _P, _P, _P, _P, _Z, _I, _I, _I, _P]),
    "lmv_block_bwd": (_I, [C.POINTER(BlockDesc), _P, _P, _P, _Z, _P, _P, _P, _P, _P, _Z, _P, _P]),
    "lmv_sstage_supported": (_I, [_I, _I, _I, _I, _I, _I, _I]),
    "lmv_sstage_wpk_bytes": (_Z, [_I, _I]),
    "lmv_sstage_vec_floats": (_Z, [_I, _I]),
    "lmv_sstage_workspace_bytes": (_Z, [_I, _I]),
    "lmv_sstage_max_images": (_I, [_I]),
    "lmv_sstage_max_concurrent": (_I, [_I]),
    "lmv_sstage_pack": (_I, [C.POINTER(SStageBlockParams), _P, _P, _P]),
    "lmv_sstage_fwd": (_I, [C.POINTER(SStageDesc), _P, _P, _P, _P, _P, _Z, _P]),
    "lmv_stem_supported": (_I, [_I, _I, _I, _I, _I]),
    "lmv_stem_wpk_bytes": (_Z, [_I, _I]),
    "lmv_stem_pack": (_I, [_P, _P, _I, _I, _I, _P, _P]),
    "lmv_stem_fwd": (_I, [_P, _I, _L, _L, _L, _L, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "lmv_stem_debug_timing": (None, [_P]),
    "lmv_stage_error_count": (_I, [_I]),
    "lmv_debug_stage_error_set": (_I, [_I]),
    "lmv_debug_launch_timing": (_I, [_I]),
    "lmv_debug_launch_timing_read": (_I, [_P, _P, _P, _P, _I]),
    "lmv_dstage_supported": (_I, [_I, _I, _I, _I, _I, _I, _I]),
    "lmv_dstage_wpk_bytes": (_Z, [_I, _I]),
    "lmv_dstage_vec_floats": (_Z, [_I, _I]),
    "lmv_dstage_workspace_bytes": (_Z, [_I, _I]),
    "lmv_dstage_max_concurrent": (_I, [_I, _I, _I]),
    "lmv_dstage_pack": (_I, [C.POINTER(DStageBlockParams), _P, _P, _P]),
    "lmv_dstage_fwd": (_I, [C.POINTER(SStageDesc), _P, _P, _P, _P, _P, _Z, _P]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"lemevit_amd: HIP kernel library not found at {LIB_PATH}. There is no CPU/PyTorch fallback; "
            "build it with `make -C lemevit_amd/csrc` (hipcc, --offload-arch=gfx950).")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise RuntimeError(f"lemevit_amd: cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = ABI mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    v = lib.lmv_abi_version()
    if v != ABI_VERSION:
        raise RuntimeError(f"lemevit_amd: ABI version mismatch (library {v}, binding {ABI_VERSION}); rebuild csrc")
    return lib


lib = _load()


def config_set(key: str, value: int) -> None:
    """lmv_config_set: change a tuning switch of the library at run time (tests, tools/ sweeps)."""
    if lib.lmv_config_set(key.encode(), int(value)):
        raise RuntimeError(f"lmv_config_set: {lib.lmv_last_error().decode(errors='replace')}")


def config_get(key: str) -> int:
    v = C.c_int(0)
    if lib.lmv_config_get(key.encode(), C.byref(v)):
        raise RuntimeError(f"lmv_config_get: {lib.lmv_last_error().decode(errors='replace')}")
    return int(v.value)


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"{what} failed (code {rc}): {lib.lmv_last_error().decode(errors='replace')}")
