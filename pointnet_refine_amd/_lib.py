"""Loader and ctypes bindings for ``libpointnet_refine_hip.so`` (C ABI declared in
``include/pointnet_refine_hip.h``).

The product path has NO CPU fallback: ``lib()`` raises if the library is missing, and
every op in ``ops.py`` raises on non-GPU tensors.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
# PRH_LIB_PATH: load an alternative build of the same library (kernel-tuning experiments)
LIB_PATH = os.environ.get("PRH_LIB_PATH") or os.path.join(_HERE, "libpointnet_refine_hip.so")
_CSRC = os.path.join(_HERE, "csrc")
# every file of csrc/ counts for staleness (prh_lib.hip is the one translation unit; it includes the rest)
_SOURCES = [os.path.join(_CSRC, "prh_lib.hip")] + (sorted(
    os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith((".hpp", ".h", ".hip")) and f != "prh_lib.hip")
    if os.path.isdir(_CSRC) else [])      # a binary-only install has no csrc/: nothing can be stale
_HEADER = os.path.join(_ROOT, "include", "pointnet_refine_hip.h")

PRH_MAX_LAYERS = 8


class BnLayer(C.Structure):
    _fields_ = [("w", C.c_void_p), ("b", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p),
                ("running_mean", C.c_void_p), ("running_var", C.c_void_p),
                ("num_batches_tracked", C.c_void_p), ("cin", C.c_int), ("cout", C.c_int)]


class BnLayerGrad(C.Structure):
    _fields_ = [("dw", C.c_void_p), ("db", C.c_void_p), ("dgamma", C.c_void_p), ("dbeta", C.c_void_p)]


class EncoderParams(C.Structure):
    _fields_ = [("in_channel", C.c_int), ("out_dim", C.c_int), ("conv", BnLayer * 5),
                ("fusion", BnLayer), ("gate_w1", C.c_void_p), ("gate_b1", C.c_void_p),
                ("gate_w2", C.c_void_p), ("gate_b2", C.c_void_p)]


class EncoderGrads(C.Structure):
    _fields_ = [("conv", BnLayerGrad * 5), ("fusion", BnLayerGrad), ("d_gate_w1", C.c_void_p),
                ("d_gate_b1", C.c_void_p), ("d_gate_w2", C.c_void_p), ("d_gate_b2", C.c_void_p)]


class EncoderSaved(C.Structure):
    _fields_ = [("z_cat", C.c_void_p), ("z_fus", C.c_void_p), ("gate", C.c_void_p),
                ("bn_scale", C.c_void_p), ("bn_shift", C.c_void_p), ("bn_mean", C.c_void_p),
                ("bn_rstd", C.c_void_p), ("argmax", C.c_void_p), ("op_amax", C.c_void_p)]


class EncoderSavedBf16(C.Structure):
    _fields_ = [("z_cat", C.c_void_p), ("z_fus", C.c_void_p), ("gate", C.c_void_p),
                ("bn_scale", C.c_void_p), ("bn_shift", C.c_void_p), ("bn_mean", C.c_void_p),
                ("bn_rstd", C.c_void_p), ("argmax", C.c_void_p)]


EXPORTS = [
    "prh_encoder_workspace_bytes", "prh_encoder_forward", "prh_encoder_backward",
    "prh_encoder_bf16_workspace_bytes", "prh_encoder_forward_bf16", "prh_encoder_backward_bf16",
    "prh_linear_bf16_workspace_bytes", "prh_linear_forward_bf16", "prh_linear_backward_bf16",
    "prh_linear_forward_out16", "prh_linear_backward_dy16", "prh_attn_forward_kv16", "prh_attn_backward_kv16",
    "prh_encoder_fused_image_bytes", "prh_encoder_fused_prepare", "prh_encoder_fused_workspace_bytes",
    "prh_encoder_fused_forward",
    "prh_linear_forward_workspace_bytes", "prh_linear_forward", "prh_linear_forward_ex",
    "prh_linear_forward_full", "prh_operand_absmax_workspace_bytes", "prh_operand_absmax",
    "prh_linear_uses_operand_maxima", "prh_linear_backward_full", "prh_pos_hidden_forward", "prh_pos_hidden_backward_workspace_bytes",
    "prh_pos_hidden_backward", "prh_linear_small_forward", "prh_linear_small_backward_workspace_bytes",
    "prh_linear_small_backward",
    "prh_linear_backward_workspace_bytes", "prh_linear_backward", "prh_linear_backward_ex",
    "prh_mlp_stack_workspace_bytes", "prh_mlp_stack_forward", "prh_mlp_stack_backward",
    "prh_test_gemm_nt", "prh_test_gemm_tn_workspace_bytes", "prh_test_gemm_tn", "prh_test_xcc_map",
    "prh_profile_enable", "prh_profile_count", "prh_profile_reset", "prh_profile_read",
    "prh_attn_forward", "prh_attn_backward", "prh_attn_backward_ex",
    "prh_context_workspace_bytes", "prh_context_build",
    "prh_l1_loss_workspace_bytes", "prh_l1_loss", "prh_adam_step",
    "prh_add_dropout_layernorm_forward", "prh_add_dropout_layernorm_workspace_bytes",
    "prh_add_dropout_layernorm_backward",
    "prh_relu_mask_absmax", "prh_cast_perm_bf16", "prh_posmem_images", "prh_attn_fold_forward", "prh_set_gemm_mode", "prh_get_gemm_mode", "prh_set_dropout_seed_source",
    "prh_last_error", "prh_version",
]

_lock = threading.Lock()
_lib = None


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.exists(s) and os.path.getmtime(s) > t for s in _SOURCES + [_HEADER])


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP library for gfx950 in-tree with hipcc (cross-compiles without a GPU)."""
    if not force and not needs_build():
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-o", LIB_PATH, _SOURCES[0]]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB_PATH


def _bind(lib):
    vp, i, f, sz, lg = C.c_void_p, C.c_int, C.c_float, C.c_size_t, C.c_long
    lib.prh_last_error.restype = C.c_char_p
    lib.prh_version.restype = C.c_char_p
    lib.prh_encoder_workspace_bytes.restype = sz
    lib.prh_encoder_workspace_bytes.argtypes = [i, i, i, i, i]
    lib.prh_encoder_forward.restype = i
    lib.prh_encoder_forward.argtypes = [C.POINTER(EncoderParams), vp, i, i, i, f, f, vp, vp,
                                        C.POINTER(EncoderSaved), vp, sz, i, vp]
    lib.prh_encoder_backward.restype = i
    lib.prh_encoder_backward.argtypes = [C.POINTER(EncoderParams), vp, i, i, i, vp, vp, i,
                                         C.POINTER(EncoderSaved), C.POINTER(EncoderGrads), vp, vp,
                                         sz, i, vp]
    lib.prh_encoder_bf16_workspace_bytes.restype = sz
    lib.prh_encoder_bf16_workspace_bytes.argtypes = [i, i, i, i, i]
    lib.prh_encoder_forward_bf16.restype = i
    lib.prh_encoder_forward_bf16.argtypes = [C.POINTER(EncoderParams), vp, i, i, i, f, f, vp, vp,
                                             C.POINTER(EncoderSavedBf16), vp, sz, i, vp]
    lib.prh_encoder_backward_bf16.restype = i
    lib.prh_encoder_backward_bf16.argtypes = [C.POINTER(EncoderParams), vp, i, i, i, vp, vp,
                                              C.POINTER(EncoderSavedBf16), C.POINTER(EncoderGrads), vp, vp,
                                              sz, i, vp]
    lib.prh_linear_bf16_workspace_bytes.restype = sz
    lib.prh_linear_bf16_workspace_bytes.argtypes = [i, i, i, i]
    lib.prh_linear_forward_bf16.restype = i
    lib.prh_linear_forward_bf16.argtypes = [vp, lg, vp, vp, vp, i, i, i, i, vp, sz, i, vp]
    lib.prh_linear_backward_bf16.restype = i
    lib.prh_linear_backward_bf16.argtypes = [vp, lg, vp, vp, vp, vp, vp, i, i, i, vp, sz, i, vp]
    lib.prh_linear_forward_out16.restype = i
    lib.prh_linear_forward_out16.argtypes = [vp, lg, vp, vp, vp, i, i, i, vp, sz, i, vp]
    lib.prh_linear_backward_dy16.restype = i
    lib.prh_linear_backward_dy16.argtypes = [vp, lg, vp, vp, vp, vp, vp, i, i, i, vp, sz, i, vp]
    lib.prh_attn_forward_kv16.restype = i
    lib.prh_attn_forward_kv16.argtypes = [vp, lg, vp, lg, vp, lg, vp, lg, vp, i, i, i, i, f, f, C.c_uint, i, vp]
    lib.prh_attn_backward_kv16.restype = i
    lib.prh_attn_backward_kv16.argtypes = [vp, lg, vp, lg, vp, lg, vp, lg, vp, vp, lg, vp, lg, vp, lg, vp, lg,
                                           i, i, i, i, f, f, C.c_uint, i, vp]
    lib.prh_encoder_fused_image_bytes.restype = sz
    lib.prh_encoder_fused_image_bytes.argtypes = [i, i]
    lib.prh_encoder_fused_prepare.restype = i
    lib.prh_encoder_fused_prepare.argtypes = [C.POINTER(EncoderParams), f, vp, vp, i, vp, sz, i, vp]
    lib.prh_encoder_fused_workspace_bytes.restype = sz
    lib.prh_encoder_fused_workspace_bytes.argtypes = [i, i, i]
    lib.prh_encoder_fused_forward.restype = i
    lib.prh_encoder_fused_forward.argtypes = [vp, i, i, i, vp, i, i, vp, vp, vp, vp, vp, sz, i, vp]
    lib.prh_linear_forward.restype = i
    lib.prh_linear_forward.argtypes = [vp, lg, vp, vp, vp, i, i, i, i, vp, sz, i, vp]
    lib.prh_linear_forward_ex.restype = i
    lib.prh_linear_forward_ex.argtypes = [vp, lg, vp, vp, vp, i, i, i, i, vp, vp, sz, i, vp]
    lib.prh_linear_forward_full.restype = i
    lib.prh_linear_forward_full.argtypes = [vp, lg, vp, vp, vp, lg, vp, i, i, i, i, vp, vp, f, C.c_uint, vp, sz, i, vp]
    lib.prh_operand_absmax_workspace_bytes.restype = sz
    lib.prh_operand_absmax_workspace_bytes.argtypes = []
    lib.prh_operand_absmax.restype = i
    lib.prh_operand_absmax.argtypes = [vp, lg, lg, i, vp, vp, sz, i, vp]
    lib.prh_linear_uses_operand_maxima.restype = i
    lib.prh_linear_uses_operand_maxima.argtypes = [i, i, i]
    lib.prh_linear_backward_full.restype = i
    lib.prh_linear_backward_full.argtypes = [vp, lg, vp, vp, vp, vp, vp, i, i, i, vp, vp, vp, vp, sz, i, vp]
    lib.prh_pos_hidden_forward.restype = i
    lib.prh_pos_hidden_forward.argtypes = [vp, lg, vp, vp, vp, lg, i, i, vp]
    lib.prh_pos_hidden_backward_workspace_bytes.restype = sz
    lib.prh_pos_hidden_backward_workspace_bytes.argtypes = [lg, i]
    lib.prh_pos_hidden_backward.restype = i
    lib.prh_pos_hidden_backward.argtypes = [vp, lg, vp, vp, vp, vp, vp, vp, lg, i, vp, sz, i, vp]
    lib.prh_linear_small_forward.restype = i
    lib.prh_linear_small_forward.argtypes = [vp, vp, vp, vp, lg, i, i, i, vp]
    lib.prh_linear_small_backward_workspace_bytes.restype = sz
    lib.prh_linear_small_backward_workspace_bytes.argtypes = [lg, i, i]
    lib.prh_linear_small_backward.restype = i
    lib.prh_linear_small_backward.argtypes = [vp, vp, vp, vp, vp, vp, lg, i, i, vp, sz, i, vp]
    lib.prh_linear_backward_ex.restype = i
    lib.prh_linear_backward_ex.argtypes = [vp, lg, vp, vp, vp, vp, vp, i, i, i, vp, vp, vp, sz, i, vp]
    lib.prh_linear_forward_workspace_bytes.restype = sz
    lib.prh_linear_forward_workspace_bytes.argtypes = [i, i, i]
    lib.prh_linear_backward_workspace_bytes.restype = sz
    lib.prh_linear_backward_workspace_bytes.argtypes = [i, i, i]
    lib.prh_linear_backward.restype = i
    lib.prh_linear_backward.argtypes = [vp, lg, vp, vp, vp, vp, vp, i, i, i, vp, sz, i, vp]
    lib.prh_mlp_stack_workspace_bytes.restype = sz
    lib.prh_mlp_stack_workspace_bytes.argtypes = [i, i, C.POINTER(BnLayer)]
    lib.prh_mlp_stack_forward.restype = i
    lib.prh_mlp_stack_forward.argtypes = [C.POINTER(BnLayer), i, i, vp, i, i, f, f, vp, vp, vp, vp,
                                          vp, vp, vp, sz, i, vp]
    lib.prh_mlp_stack_backward.restype = i
    lib.prh_mlp_stack_backward.argtypes = [C.POINTER(BnLayer), i, i, vp, i, i, vp, vp, vp, vp, vp,
                                           vp, C.POINTER(BnLayerGrad), vp, vp, sz, i, vp]
    lib.prh_attn_forward.restype = i
    lib.prh_attn_forward.argtypes = [vp, lg, vp, lg, vp, lg, vp, lg, vp, i, i, i, i, f, f, C.c_uint, i, vp]
    lib.prh_attn_backward.restype = i
    lib.prh_attn_backward.argtypes = [vp, lg, vp, lg, vp, lg, vp, lg, vp, vp, lg, vp, lg, vp, lg, vp, lg,
                                      i, i, i, i, f, f, C.c_uint, i, vp]
    lib.prh_context_workspace_bytes.restype = C.c_size_t
    lib.prh_context_workspace_bytes.argtypes = [i, i, i]
    lib.prh_context_build.restype = i
    lib.prh_context_build.argtypes = [vp, i, vp, i, vp, i, i, f, f, i, i, C.c_ulonglong, vp, vp, vp, vp,
                                      C.c_size_t, i, vp]
    lib.prh_add_dropout_layernorm_forward.restype = i
    lib.prh_add_dropout_layernorm_forward.argtypes = [vp, vp, vp, vp, lg, i, f, f, C.c_uint, vp, vp, vp, i, vp]
    lib.prh_add_dropout_layernorm_workspace_bytes.restype = C.c_size_t
    lib.prh_add_dropout_layernorm_workspace_bytes.argtypes = []
    lib.prh_add_dropout_layernorm_backward.restype = i
    lib.prh_add_dropout_layernorm_backward.argtypes = [vp, vp, vp, vp, vp, vp, lg, i, f, C.c_uint, vp, vp, vp, vp,
                                                       vp, C.c_size_t, i, vp]
    lib.prh_l1_loss_workspace_bytes.restype = C.c_size_t
    lib.prh_l1_loss_workspace_bytes.argtypes = []
    lib.prh_l1_loss.restype = i
    lib.prh_l1_loss.argtypes = [vp, vp, i, lg, C.c_double, i, vp, vp, vp, C.c_double, vp, C.c_size_t, i, vp]
    lib.prh_adam_step.restype = i
    lib.prh_adam_step.argtypes = [vp, vp, vp, vp, lg, f, f, f, f, f, i, i, vp]
    lib.prh_cast_perm_bf16.restype = i
    lib.prh_cast_perm_bf16.argtypes = [vp, lg, vp, lg, i, vp]
    lib.prh_posmem_images.restype = i
    lib.prh_posmem_images.argtypes = [vp, lg, vp, vp, vp, vp, vp, lg, lg, vp, vp, i, vp]
    lib.prh_attn_fold_forward.restype = i
    lib.prh_attn_fold_forward.argtypes = [vp, lg, vp, vp, vp, lg, vp, lg, vp, vp, lg, i, i, i, i, C.c_float, i, vp]
    lib.prh_relu_mask_absmax.restype = i
    lib.prh_relu_mask_absmax.argtypes = [vp, vp, vp, lg, f, vp, vp, sz, i, vp]
    lib.prh_set_gemm_mode.restype = i
    lib.prh_set_gemm_mode.argtypes = [i]
    lib.prh_attn_backward_ex.restype = i
    lib.prh_attn_backward_ex.argtypes = [vp, lg, vp, lg, vp, lg, vp, lg, vp, vp, lg, vp, lg, vp, lg, vp, lg,
                                         i, i, i, i, f, f, C.c_uint, vp, i, vp]
    lib.prh_set_dropout_seed_source.restype = i
    lib.prh_set_dropout_seed_source.argtypes = [vp]
    lib.prh_get_gemm_mode.restype = i
    lib.prh_profile_enable.restype = i
    lib.prh_profile_enable.argtypes = [i]
    lib.prh_profile_count.restype = i
    lib.prh_profile_reset.restype = i
    lib.prh_profile_read.restype = i
    lib.prh_profile_read.argtypes = [i, C.c_char_p, i, C.POINTER(C.c_float), C.POINTER(C.c_double),
                                     C.POINTER(C.c_double)]
    lib.prh_test_gemm_nt.restype = i
    lib.prh_test_gemm_nt.argtypes = [vp, vp, vp, i, i, i, vp, sz, i, vp]
    lib.prh_test_gemm_tn_workspace_bytes.restype = sz
    lib.prh_test_gemm_tn_workspace_bytes.argtypes = [i, i, i]
    lib.prh_test_gemm_tn.restype = i
    lib.prh_test_gemm_tn.argtypes = [vp, vp, vp, vp, i, i, i, vp, sz, i, vp]
    lib.prh_test_xcc_map.restype = i
    lib.prh_test_xcc_map.argtypes = [i, i, vp, i, vp]
    return lib


def lib():
    """The loaded library; raises RuntimeError if it has not been built."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise RuntimeError(
                        f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                        "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
                _lib = _bind(C.CDLL(LIB_PATH))
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().prh_last_error().decode(errors="replace")
        if rc == -1:
            raise RuntimeError(f"{what}: {msg}")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")
