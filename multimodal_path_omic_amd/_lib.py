"""ctypes binding of libmpo_hip.so (C ABI: include/mpo_hip.h).

There is deliberately no fallback: if the shared library is missing, or a call returns
non-zero, a RuntimeError is raised.  Tensors cross the boundary as raw device pointers
(`tensor.data_ptr()`), sizes, and the current HIP stream of the tensor's device.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_size_t, c_uint64, c_void_p  # noqa: F401

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmpo_hip.so")

MPO_F32, MPO_BF16 = 0, 1


class BagPlanC(ctypes.Structure):
    """mpo_bag_plan of include/mpo_hip.h (host struct; wg_start is a device pointer)."""
    _fields_ = [("wg_start", c_void_p), ("n_wg", ctypes.c_int32), ("rows_per_wg", ctypes.c_int32)]
ACT = {"none": 0, "relu": 1, "elu": 2, "tanh": 3, "sigmoid": 4}

_lib = None

_P = c_void_p
_SIGNATURES = {
    "mpo_abi_version": (c_int, []),
    "mpo_last_error": (c_char_p, []),
    "mpo_linear_forward": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_float, c_int, _P]),
    "mpo_linear_backward_input": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_float, c_int, _P]),
    "mpo_linear_backward_weight": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_float, _P]),
    "mpo_coattn_splits": (c_int, [c_int, c_int]),
    "mpo_coattn_target_workgroups": (c_int, []),
    "mpo_map_block_dot": (c_int, [_P, _P, _P, c_int, c_int, _P, _P]),
    "mpo_map_block_scale": (c_int, [_P, _P, _P, c_int, c_int, _P, _P]),
    "mpo_key_projection": (c_int, [_P, ctypes.c_int64, c_int, _P, _P, _P, _P]),
    "mpo_nacagat_fwd_bagpass": (c_int, [_P, _P, c_int, c_int, _P, _P, _P, _P, c_int, c_int, _P, _P]),
    "mpo_coattn_fwd_bagpass": (c_int, [_P, c_int, _P, c_int, c_int, _P, _P, _P, _P, c_int, c_int, _P, _P]),
    "mpo_coattn_bwd_bagpass": (c_int, [_P, c_int, _P, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, _P, _P]),
    "mpo_coattn_saved_floats": (c_size_t, [c_int, c_int, c_int]),
    "mpo_coattn_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "mpo_coattn_mcat_forward": (c_int, [_P, c_int, _P, c_int, c_int, c_int, _P, c_int, c_int, _P, _P, _P, _P,
                                        _P, _P, _P, _P, _P, c_size_t, _P]),
    "mpo_patch_coattn_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "mpo_patch_coattn_mcat_forward": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, _P, c_float, c_uint64, c_uint64, _P,
                                              _P, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "mpo_patch_coattn_fwd_bagpass": (c_int, [_P, _P, _P, _P, c_int, _P, _P, _P, _P, c_int, c_int, c_float, c_uint64, c_uint64,
                                             _P, _P]),
    "mpo_pack_patch_weight": (c_int, [_P, _P, c_int, c_int, _P]),
    "mpo_patch_fc_workspace_bytes": (c_size_t, [c_int, c_int]),
    "mpo_patch_fc_f32_workspace_bytes": (c_size_t, [c_int]),
    "mpo_patch_fc_f32_forward": (c_int, [_P, ctypes.c_int64, c_int, _P, _P, c_int, c_float, c_uint64, c_uint64, _P, c_float, _P, _P, c_size_t, _P]),
    "mpo_patch_fc_f32_backward": (c_int, [_P, _P, _P, ctypes.c_int64, c_int, c_int, c_float, _P, _P, _P, c_size_t, _P]),
    "mpo_patch_fc_forward": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, _P, c_int, c_float, c_uint64, c_uint64, _P, _P, _P,
                                     _P, c_size_t, _P]),
    "mpo_coattn_mcat_backward": (c_int, [_P, c_int, _P, c_int, c_int, c_int, _P, c_int, c_int, _P, _P, _P, _P,
                                         _P, _P, _P, c_int, _P, _P, _P, _P, _P, _P, c_float, _P, _P, c_size_t, _P]),
    "mpo_colsum_bf16": (c_int, [_P, _P, ctypes.c_int64, c_int, _P]),
    "mpo_patch_weight_grad_workspace_bytes": (c_size_t, [c_int, c_int]),
    "mpo_patch_weight_grad": (c_int, [_P, _P, ctypes.c_int64, c_int, c_int, _P, c_int, _P, c_size_t, _P]),
    "mpo_adam_step_flat": (c_int, [_P, _P, _P, _P, ctypes.c_int64, c_float, c_float, c_float, c_float, c_float, c_int, _P, _P]),
    "mpo_patch_epilogue_forward": (c_int, [_P, _P, ctypes.c_int64, c_int, c_float, c_uint64, c_uint64, _P, _P]),
    "mpo_patch_epilogue_backward_workspace_bytes": (c_size_t, [ctypes.c_int64, c_int]),
    "mpo_patch_epilogue_backward": (c_int, [_P, _P, _P, ctypes.c_int64, c_int, c_float, _P, _P, c_size_t, _P]),
    "mpo_nacagat_saved_floats": (c_size_t, [c_int, c_int, c_int]),
    "mpo_nacagat_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "mpo_coattn_nacagat_forward": (c_int, [_P, c_int, _P, c_int, _P, c_int, c_int, c_int, _P, c_int, c_int, _P, _P, _P, _P,
                                           c_float, c_uint64, c_uint64, _P, _P, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "mpo_coattn_nacagat_backward": (c_int, [_P, c_int, _P, c_int, _P, c_int, c_int, c_int, _P, c_int, c_int, _P, _P, _P,
                                            c_float, c_uint64, c_uint64, _P, _P, _P, _P, _P, _P, _P,
                                            _P, c_int, _P, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "mpo_nacagat_patch_grad": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, c_float, _P, _P, _P, c_size_t, _P]),
    "mpo_nacagat_patch_grad_fused": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, c_float, _P, _P, _P,
                                             c_size_t, _P]),
    "mpo_survival_head_forward": (c_int, [_P, c_int, c_int, _P, _P, _P, _P]),
    "mpo_survival_head_backward": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, _P, _P]),
    "mpo_ces_loss_forward": (c_int, [_P, _P, _P, _P, c_int, c_int, c_float, c_float, _P, _P, _P]),
    "mpo_ces_loss_backward": (c_int, [_P, _P, _P, _P, c_int, c_int, c_float, c_float, _P, c_int, _P, _P, _P]),
    "mpo_encoder_saved_floats": (c_size_t, [c_int] * 6),
    "mpo_encoder_workspace_bytes": (c_size_t, [c_int] * 4),
    "mpo_encoder_rng_span": (c_uint64, [c_int] * 5),
    "mpo_encoder_forward": (c_int, [_P] + [c_int] * 7 + [_P, c_float, c_uint64, c_uint64, _P, _P, _P, _P]),
    "mpo_encoder_backward": (c_int, [_P] + [c_int] * 7 + [_P, c_float, c_uint64, c_uint64, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "mpo_bag_self_attention_saved_floats": (c_size_t, [c_int] * 4),
    "mpo_bag_self_attention_workspace_bytes": (c_size_t, [c_int] * 4),
    "mpo_set_bag_self_attention_bf16x3": (c_int, [c_int]),
    "mpo_bag_self_attention_forward": (c_int, [_P, c_int, c_int, c_int, c_int, c_float, c_uint64, c_uint64, _P, _P, _P, _P, _P]),
    "mpo_bag_self_attention_backward": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, c_uint64, c_uint64, _P, _P, _P, c_size_t, _P]),
    "mpo_gated_pool_saved_floats": (c_size_t, [c_int] * 3),
    "mpo_gated_pool_workspace_bytes": (c_size_t, [c_int] * 3),
    "mpo_gated_pool_rng_span": (c_uint64, [c_int] * 3),
    "mpo_gated_pool_forward": (c_int, [_P, c_int, c_int, c_int, c_int, _P, c_float, c_float, c_uint64, c_uint64, _P, _P, _P, c_int, _P, _P]),
    "mpo_gated_pool_backward": (c_int, [_P, c_int, c_int, c_int, c_int, _P, c_float, c_float, _P, _P, _P, c_int, _P, _P, _P, _P, c_size_t, _P]),
    "mpo_fusion_head_saved_floats": (c_size_t, [c_int] * 4),
    "mpo_fusion_head_workspace_bytes": (c_size_t, [c_int] * 4),
    "mpo_fusion_head_forward": (c_int, [_P] + [c_int] * 5 + [_P, _P, _P, _P, _P, _P]),
    "mpo_fusion_head_backward": (c_int, [_P] + [c_int] * 5 + [_P] * 10 + [_P, c_size_t, _P]),
    "mpo_fusion_head_loss_saved_floats": (c_size_t, [c_int] * 4),
    "mpo_fusion_head_loss_forward": (c_int, [_P] + [c_int] * 5 + [_P, _P, _P, _P, c_float, c_float] + [_P] * 6 + [_P]),
    "mpo_fusion_head_loss_backward": (c_int, [_P] + [c_int] * 5 + [_P, _P, _P, _P] + [_P, c_size_t, _P]),
    "mpo_step_counters_bump": (c_int, [_P, _P, _P]),
    "mpo_set_gemm_fast_path": (c_int, [c_int]),
    "mpo_set_coattn_bwd_two_wave": (c_int, [c_int]),
    "mpo_set_nacagat_one_pass_key_grad": (c_int, [c_int]),
    "mpo_set_coattn_bwd_f32_vector": (c_int, [c_int]),
    "mpo_omic_snn_saved_floats": (c_size_t, [c_int] * 3),
    "mpo_omic_snn_workspace_bytes": (c_size_t, [c_int] * 3),
    "mpo_omic_snn_rng_span": (c_uint64, [c_int] * 3),
    "mpo_omic_snn_forward": (c_int, [_P, _P, c_int, c_int, c_int, _P, c_float, c_uint64, c_uint64, _P, _P, _P, _P]),
    "mpo_omic_snn_backward": (c_int, [_P, _P, c_int, c_int, c_int, _P, c_float, c_uint64, c_uint64, _P, _P, _P, _P, _P,
                                      _P, c_size_t, _P]),
    "mpo_cag_saved_floats": (c_size_t, [c_int] * 2),
    "mpo_cag_workspace_bytes": (c_size_t, [c_int] * 2),
    "mpo_cag_forward": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P]),
    "mpo_cag_backward": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, _P, _P, _P, c_int, _P, _P, _P, c_size_t, _P]),
}


def exported_symbols():
    """Names every build of the library must export (checked by the CPU test-suite)."""
    return list(_SIGNATURES)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension has not been built "
                "(run `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().mpo_last_error()
        raise RuntimeError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  The tensor must be contiguous and on the GPU."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("libmpo_hip operates on GPU tensors only (got a CPU tensor); there is no CPU fallback")
    if not t.is_contiguous():
        raise RuntimeError("libmpo_hip needs contiguous tensors")
    return t.data_ptr()


def ptr_array(tensors):
    """ctypes array of device pointers (kept alive by the caller for the duration of the call)."""
    arr = (c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = ptr(t)
    return arr


def stream_of(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def bag_dtype_code(t):
    if t.dtype == torch.float32:
        return MPO_F32
    if t.dtype == torch.bfloat16:
        return MPO_BF16
    raise RuntimeError(f"bag dtype must be float32 or bfloat16, got {t.dtype}")
