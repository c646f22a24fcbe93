"""ctypes binding of libaa_mi355.so (C ABI: include/aa_mi355.h).

The library is the only compute backend of this package: if it cannot be loaded the import of any
op fails loudly (RuntimeError) -- there is no eager/PyTorch fallback.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must precede the dlopen below: libaa_mi355.so has to bind to the HIP runtime
#                           torch already loaded, otherwise two runtimes end up in one process)

AA_F16, AA_BF16, AA_F32 = 0, 1, 2
AA_ACT_NONE, AA_ACT_SILU, AA_ACT_GELU, AA_ACT_QUICK_GELU = 0, 1, 2, 3


class AaConvGemm(C.Structure):
    _fields_ = [
        ("a0", C.c_void_p), ("a1", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p),
        ("rowvec", C.c_void_p), ("residual", C.c_void_p), ("out", C.c_void_p),
        ("c0", C.c_int32), ("c1", C.c_int32),
        ("n_img", C.c_int32), ("h_in", C.c_int32), ("w_in", C.c_int32), ("h_virt", C.c_int32),
        ("w_virt", C.c_int32), ("h_out", C.c_int32), ("w_out", C.c_int32),
        ("kh", C.c_int32), ("kw", C.c_int32), ("stride", C.c_int32), ("pad_h", C.c_int32), ("pad_w", C.c_int32),
        ("n_out", C.c_int32), ("n_pad", C.c_int32), ("k_pad", C.c_int32),
        ("rowvec_div", C.c_int32), ("ldo", C.c_int32), ("ldr", C.c_int32),
        ("act", C.c_int32), ("geglu", C.c_int32), ("bias_per_row", C.c_int32),
        ("dtype", C.c_int32), ("out_dtype", C.c_int32), ("out_scale", C.c_float), ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64), ("k_order", C.c_int32), ("debug", C.c_int32), ("tile", C.c_int32), ("k_splits", C.c_int32), ("rowvec_ld", C.c_int32), ("acc_scale", C.c_float),
        ("out_sy", C.c_int32), ("out_sx", C.c_int32), ("out_oy", C.c_int32), ("out_ox", C.c_int32),
        ("ln_stats", C.c_void_p), ("ln_cols", C.c_void_p), ("ln_parts", C.c_int32), ("ln_eps", C.c_float),
        ("row_stats", C.c_void_p), ("row_stats_parts", C.c_int32), ("tickets_len", C.c_int32), ("tickets", C.c_void_p),
        ("row_coef", C.c_void_p), ("row_coef_eps", C.c_float), ("_reserved107", C.c_int32),
    ]


class AaGroupNorm(C.Structure):
    _fields_ = [
        ("x0", C.c_void_p), ("x1", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p), ("y", C.c_void_p),
        ("c0", C.c_int32), ("c1", C.c_int32), ("n_groups_img", C.c_int32), ("tokens_per_group", C.c_int32),
        ("num_groups", C.c_int32), ("silu", C.c_int32), ("dtype", C.c_int32), ("eps", C.c_float),
    ]


class AaAttnOperand(C.Structure):
    _fields_ = [
        ("ptr", C.c_void_p), ("outer_stride", C.c_int64), ("inner_stride", C.c_int64), ("pos_stride", C.c_int64),
        ("ld", C.c_int32), ("col0", C.c_int32), ("outer_div", C.c_int32), ("seq_mod", C.c_int32),
    ]


class AaAttention(C.Structure):
    _fields_ = [
        ("q", AaAttnOperand), ("k", AaAttnOperand), ("v", AaAttnOperand), ("o", AaAttnOperand),
        ("n_outer", C.c_int32), ("n_inner", C.c_int32), ("heads", C.c_int32), ("head_dim", C.c_int32),
        ("q_len", C.c_int32), ("kv_len", C.c_int32), ("dtype", C.c_int32), ("scale", C.c_float), ("causal", C.c_int32), ("_pad", C.c_int32),
    ]


class AaSeqSelfAttn(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("w", C.c_void_p), ("w_bias", C.c_void_p), ("pre_w", C.c_void_p), ("pre_bias", C.c_void_p), ("pre_residual", C.c_void_p), ("pre_out", C.c_void_p), ("o", C.c_void_p),
        ("outer_stride", C.c_int64), ("inner_stride", C.c_int64), ("pos_stride", C.c_int64), ("x_bytes", C.c_int64), ("o_bytes", C.c_int64),
        ("n_outer", C.c_int32), ("n_inner", C.c_int32), ("seq_len", C.c_int32), ("channels", C.c_int32), ("ldx", C.c_int32), ("ldo", C.c_int32), ("ld_res", C.c_int32), ("ld_pre", C.c_int32),
        ("normalize", C.c_int32), ("ln_eps", C.c_float), ("scale", C.c_float), ("dtype", C.c_int32), ("flags", C.c_int32),
    ]


class AaFFFused(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("outer", C.c_void_p), ("out", C.c_void_p), ("w", C.c_void_p), ("rows", C.c_int64),
        ("channels", C.c_int32), ("ldx", C.c_int32), ("ld_outer", C.c_int32), ("ldo", C.c_int32),
        ("normalize", C.c_int32), ("ln_eps", C.c_float), ("dtype", C.c_int32), ("flags", C.c_int32),
    ]


class AaLinearRows(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("residual", C.c_void_p), ("out", C.c_void_p), ("w", C.c_void_p), ("rows", C.c_int64),
        ("channels", C.c_int32), ("n_out", C.c_int32), ("ldx", C.c_int32), ("ld_res", C.c_int32), ("ldo", C.c_int32),
        ("normalize", C.c_int32), ("ln_eps", C.c_float), ("dtype", C.c_int32), ("flags", C.c_int32),
        ("row_affine", C.c_void_p), ("rows_per_group", C.c_int32), ("_pad", C.c_int32),
    ]


class AaDpmStep(C.Structure):
    _fields_ = [
        ("eps_uncond", C.c_void_p), ("eps_text", C.c_void_p), ("latents", C.c_void_p), ("x0_prev", C.c_void_p),
        ("latents_lp", C.c_void_p), ("n", C.c_int64),
        ("guidance", C.c_float), ("sigma_s", C.c_float), ("alpha_s", C.c_float),
        ("c_x", C.c_float), ("c_d0", C.c_float), ("c_d1", C.c_float), ("dtype", C.c_int32),
    ]


class AaPackLatents(C.Structure):
    _fields_ = [
        ("sample", C.c_void_p), ("cond", C.c_void_p), ("mask", C.c_void_p), ("out", C.c_void_p),
        ("batch", C.c_int32), ("sample_batch", C.c_int32), ("cond_batch", C.c_int32), ("mask_batch", C.c_int32),
        ("channels", C.c_int32), ("frames", C.c_int32), ("hw", C.c_int32), ("dtype", C.c_int32), ("sample_dtype", C.c_int32),
    ]


class AaDpmStepTok(C.Structure):
    _fields_ = [
        ("eps_tokens", C.c_void_p), ("latents", C.c_void_p), ("x0_prev", C.c_void_p), ("latents_lp", C.c_void_p),
        ("next_t", C.c_void_p), ("next_t_count", C.c_int32), ("next_t_value", C.c_float),
        ("clips", C.c_int32), ("channels", C.c_int32), ("frames", C.c_int32), ("hw", C.c_int32),
        ("eps_ld", C.c_int32), ("guidance_on", C.c_int32),
        ("guidance", C.c_float), ("sigma_s", C.c_float), ("alpha_s", C.c_float),
        ("c_x", C.c_float), ("c_d0", C.c_float), ("c_d1", C.c_float), ("dtype", C.c_int32),
    ]


class AaBlend(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("y", C.c_void_p), ("rowvec", C.c_void_p), ("out", C.c_void_p), ("rows", C.c_int64),
        ("channels", C.c_int32), ("rowvec_div", C.c_int32), ("rowvec_mod", C.c_int32), ("rowvec_ld", C.c_int32),
        ("a", C.c_float), ("b", C.c_float), ("act", C.c_int32), ("dtype", C.c_int32),
    ]


class AaPackFrames(C.Structure):
    _fields_ = [
        ("src", C.c_void_p * 3), ("src_channels", C.c_int32 * 3), ("src_batch", C.c_int32 * 3), ("src_f32", C.c_int32 * 3),
        ("scale", C.c_void_p), ("scaled_src", C.c_int32), ("out", C.c_void_p),
        ("batch", C.c_int32), ("frames", C.c_int32), ("hw", C.c_int32), ("out_channels", C.c_int32), ("dtype", C.c_int32),
    ]


class AaEulerStepTok(C.Structure):
    _fields_ = [
        ("v_tokens", C.c_void_p), ("latents", C.c_void_p), ("guidance", C.c_void_p), ("next_t", C.c_void_p),
        ("next_t_count", C.c_int32), ("next_t_value", C.c_float), ("next_scale", C.c_void_p), ("next_scale_value", C.c_float),
        ("clips", C.c_int32), ("channels", C.c_int32), ("frames", C.c_int32), ("hw", C.c_int32), ("ld", C.c_int32),
        ("c_x", C.c_float), ("c_v", C.c_float), ("dtype", C.c_int32),
    ]


SYMBOLS = ("aa_version", "aa_last_error", "aa_set_tile_override", "aa_conv_gemm_tile_info", "aa_conv_gemm_tile_ok", "aa_conv_gemm_workspace", "aa_conv_gemm", "aa_conv_gemm_launch_count", "aa_conv_gemm_row_stats_parts", "aa_conv_gemm_row_coef_ok", "aa_conv_gemm_tickets", "aa_conv_gemm_reduce_launches", "aa_conv_gemm_tile_flags", "aa_ln_finalize", "aa_groupnorm_workspace", "aa_groupnorm", "aa_groupnorm_coef", "aa_set_groupnorm_two_pass", "aa_groupnorm_plan",
           "aa_layernorm", "aa_attention", "aa_seq_self_attention_ok", "aa_seq_self_attention", "aa_ff_fused_ok", "aa_ff_fused", "aa_linear_rows_ok", "aa_linear_rows", "aa_softmax_rows", "aa_cfg_dpm_step",
           "aa_timestep_embedding", "aa_pack_latents", "aa_cfg_dpm_step_tokens",
           "aa_blend", "aa_pack_frames", "aa_cfg_euler_step_tokens")

DEFAULT_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libaa_mi355.so")


def bind(path: str) -> C.CDLL:
    """dlopen `path` and attach the prototypes of include/aa_mi355.h."""
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: build it with `python -m animate_anything_amd.build` "
                           "(hipcc --offload-arch=gfx950); this package has no fallback backend")
    lib = C.CDLL(path)
    for s in SYMBOLS:
        if not hasattr(lib, s):
            raise RuntimeError(f"{path} does not export {s}")
    lib.aa_version.restype = C.c_int
    lib.aa_last_error.restype = C.c_char_p
    lib.aa_set_tile_override.argtypes = [C.c_int]
    lib.aa_set_tile_override.restype = None
    lib.aa_conv_gemm_tile_info.argtypes = [C.c_int, C.POINTER(C.c_int32)]
    lib.aa_conv_gemm_tile_info.restype = C.c_int
    lib.aa_conv_gemm_tile_ok.argtypes = [C.POINTER(AaConvGemm), C.c_int]
    lib.aa_conv_gemm_tile_ok.restype = C.c_int
    lib.aa_conv_gemm.argtypes = [C.POINTER(AaConvGemm), C.c_void_p]
    lib.aa_ln_finalize.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_void_p]
    lib.aa_ln_finalize.restype = C.c_int
    lib.aa_conv_gemm_row_stats_parts.argtypes = [C.POINTER(AaConvGemm)]
    lib.aa_conv_gemm_row_stats_parts.restype = C.c_int
    lib.aa_conv_gemm_row_coef_ok.argtypes = [C.POINTER(AaConvGemm)]
    lib.aa_conv_gemm_row_coef_ok.restype = C.c_int
    lib.aa_conv_gemm_launch_count.argtypes = [C.POINTER(AaConvGemm)]
    lib.aa_conv_gemm_tickets.argtypes = [C.POINTER(AaConvGemm)]
    lib.aa_conv_gemm_reduce_launches.argtypes = [C.POINTER(AaConvGemm)]
    lib.aa_conv_gemm_tile_flags.argtypes = [C.c_int]
    lib.aa_set_groupnorm_two_pass.argtypes = [C.c_int]
    lib.aa_set_groupnorm_two_pass.restype = None
    lib.aa_groupnorm_plan.argtypes = [C.POINTER(AaGroupNorm), C.POINTER(C.c_int32)]
    lib.aa_groupnorm_plan.restype = C.c_int
    lib.aa_conv_gemm_workspace.argtypes = [C.POINTER(AaConvGemm)]
    lib.aa_conv_gemm_workspace.restype = C.c_size_t
    lib.aa_groupnorm_workspace.argtypes = [C.POINTER(AaGroupNorm)]
    lib.aa_groupnorm_workspace.restype = C.c_size_t
    lib.aa_groupnorm.argtypes = [C.POINTER(AaGroupNorm), C.c_void_p, C.c_size_t, C.c_void_p]
    lib.aa_layernorm.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32,
                                 C.c_float, C.c_int32, C.c_void_p]
    lib.aa_attention.argtypes = [C.POINTER(AaAttention), C.c_void_p]
    lib.aa_seq_self_attention_ok.argtypes = [C.POINTER(AaSeqSelfAttn)]
    lib.aa_seq_self_attention.argtypes = [C.POINTER(AaSeqSelfAttn), C.c_void_p]
    lib.aa_ff_fused_ok.argtypes = [C.POINTER(AaFFFused)]
    lib.aa_ff_fused.argtypes = [C.POINTER(AaFFFused), C.c_void_p]
    lib.aa_groupnorm_coef.argtypes = [C.POINTER(AaGroupNorm), C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    lib.aa_linear_rows_ok.argtypes = [C.POINTER(AaLinearRows)]
    lib.aa_linear_rows.argtypes = [C.POINTER(AaLinearRows), C.c_void_p]
    lib.aa_softmax_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    lib.aa_cfg_dpm_step.argtypes = [C.POINTER(AaDpmStep), C.c_void_p]
    lib.aa_timestep_embedding.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    lib.aa_pack_latents.argtypes = [C.POINTER(AaPackLatents), C.c_void_p]
    lib.aa_cfg_dpm_step_tokens.argtypes = [C.POINTER(AaDpmStepTok), C.c_void_p]
    lib.aa_blend.argtypes = [C.POINTER(AaBlend), C.c_void_p]
    lib.aa_pack_frames.argtypes = [C.POINTER(AaPackFrames), C.c_void_p]
    lib.aa_cfg_euler_step_tokens.argtypes = [C.POINTER(AaEulerStepTok), C.c_void_p]
    for s in SYMBOLS[5:]:
        if s not in ("aa_groupnorm_workspace", "aa_conv_gemm_workspace"):
            getattr(lib, s).restype = C.c_int
    return lib


_lib = None
_host_pointers_ok = False     # only the emulator build (tests/emu) accepts host memory


def get() -> C.CDLL:
    global _lib
    if _lib is None:
        _lib = bind(os.environ.get("AA_LIBRARY") or DEFAULT_PATH)      # AA_LIBRARY: another BUILD of this library (A/B and probe variants of build.py)
    return _lib


def host_pointers_ok() -> bool:
    return _host_pointers_ok


class use_library:
    """Test hook: route ops through another build of the same C ABI (the CPU SIMT emulator of
    tests/emu).  Never used by the product path."""

    def __init__(self, lib, host_pointers=False):
        self.lib, self.host = lib, host_pointers

    def __enter__(self):
        global _lib, _host_pointers_ok
        self.prev = (_lib, _host_pointers_ok)
        _lib, _host_pointers_ok = self.lib, self.host
        return self.lib

    def __exit__(self, *a):
        global _lib, _host_pointers_ok
        _lib, _host_pointers_ok = self.prev
