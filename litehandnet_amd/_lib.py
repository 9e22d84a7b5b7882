"""ctypes binding of liblhn.so -- the C ABI declared in include/lhn.h.

There is no CPU fallback: if the shared library is missing or no gfx950 device is usable the
product path raises.  (The CPU oracle lives in /oracle and is test infrastructure only.)
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblhn.so")
_lib = None


class LhnError(RuntimeError):
    pass


class View(C.Structure):
    _fields_ = [("data", C.c_void_p), ("table", C.c_void_p), ("gate", C.c_void_p),
                ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("cstride", C.c_int32), ("coff", C.c_int32), ("C", C.c_int32), ("pend", C.c_void_p)]


class GradView(C.Structure):
    _fields_ = [("dz", C.c_void_p), ("dpool", C.c_void_p), ("coef", C.c_void_p)]


class Op(C.Structure):
    _fields_ = [("kind", C.c_int32),
                ("in_buf", C.c_int32 * 3), ("in_coff", C.c_int32 * 3), ("in_C", C.c_int32 * 3), ("reserved", C.c_int32 * 6),
                ("out_buf", C.c_int32), ("out_coff", C.c_int32), ("out_C", C.c_int32),
                ("p", C.c_int32 * 12), ("ws", C.c_int64 * 12), ("i", C.c_int32 * 8), ("f", C.c_float * 8)]


class Buf(C.Structure):
    _fields_ = [("data_off", C.c_int64), ("table_off", C.c_int64), ("gate_off", C.c_int64),
                ("grad_off", C.c_int64), ("dpool_off", C.c_int64), ("coef_off", C.c_int64),
                ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32)]


# every exported symbol of include/lhn.h (tests check the .so exports all of them)
SYMBOLS = [
    "lhn_version", "lhn_deterministic", "lhn_last_error", "lhn_device_ok",
    "lhn_heatmap_encode", "lhn_heatmap_argmax", "lhn_heatmap_refine", "lhn_transform_preds",
    "lhn_heatmap_decode", "lhn_heatmap_decode_dark", "lhn_heatmap_decode_dark_udp", "lhn_heatmap_nms", "lhn_heatmap_topk", "lhn_pck_accuracy",
    "lhn_loss_balanced_mse_fwd", "lhn_loss_balanced_mse_bwd", "lhn_affine_warp_normalize", "lhn_affine_warp_normalize2", "lhn_random_flip", "lhn_hsv_jitter", "lhn_simdr_encode", "lhn_simdr_loss_fwd", "lhn_simdr_loss_bwd",
    "lhn_conv_pw_fwd", "lhn_conv_pw_fwd2", "lhn_conv_pw_bwd2", "lhn_conv_dw_fwd", "lhn_conv_dw_fwd2", "lhn_conv_dw_fwd3", "lhn_conv_stem_fwd", "lhn_conv_kxk_fwd",
    "lhn_bn_finalize", "lhn_table_fill", "lhn_table_bias", "lhn_fold_bn", "lhn_ew_fwd", "lhn_ew_fwd2", "lhn_ew_fwd3", "lhn_ew_mul_bwd", "lhn_bilinear_bwd", "lhn_shuffle2_fwd", "lhn_shuffle2_bwd", "lhn_bn_finalize2", "lhn_bn_bwd_finalize2", "lhn_maxpool2_fwd", "lhn_avgpool_fwd", "lhn_avgpool_fwd2", "lhn_avgpool_bwd2", "lhn_se_mlp_fwd2", "lhn_se_mlp_bwd2", "lhn_ca_mlp_fwd", "lhn_att_mlp_fwd", "lhn_att_mlp_bwd", "lhn_se_mlp_fwd", "lhn_se_mlp_bwd",
    "lhn_bn_bwd_reduce", "lhn_bn_bwd_finalize", "lhn_conv_pw_bwd", "lhn_conv_dw_bwd", "lhn_conv_dw_bwd2", "lhn_conv_dw_bwd3", "lhn_conv_stem_bwd",
    "lhn_conv_kxk_bwd", "lhn_ew_bwd", "lhn_ew_bwd2", "lhn_maxpool2_bwd", "lhn_maxpool2_bwd2", "lhn_maxpool2_bwd3", "lhn_ew_bwd3", "lhn_ew_bwd_multi", "lhn_avgpool_bwd3", "lhn_conv_pw_bwd3", "lhn_avgpool_bwd", "lhn_gate_bwd_reduce", "lhn_gate_bwd_reduce2", "lhn_gate_bwd_reduce3", "lhn_adam_step", "lhn_avgpool_fwd3", "lhn_avgpool_fwd4", "lhn_ca_mlp_bwd2",
    "lhn_ca_mlp_bwd", "lhn_reduce_replicas", "lhn_fold_stat_replicas", "lhn_plan_create", "lhn_plan_destroy", "lhn_plan_run", "lhn_plan_run_range",
]


def lib():
    """Load liblhn.so (once).  Raises LhnError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LhnError(f"{LIB_PATH} is missing: run `python -m litehandnet_amd.build` "
                           "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        _lib = C.CDLL(LIB_PATH)
        _lib.lhn_last_error.restype = C.c_char_p
        _lib.lhn_plan_create.restype = C.c_void_p
        _lib.lhn_plan_destroy.restype = None
    return _lib


def check(status, what=""):
    if status != 0:
        raise LhnError(f"{what}: status {status}: {lib().lhn_last_error().decode()}")


def require_device(t=None):
    if not torch.cuda.is_available():
        raise LhnError("no GPU visible: litehandnet_amd has no CPU fallback")
    if t is not None and not t.is_cuda:
        raise LhnError("tensor must live on the GPU (litehandnet_amd has no CPU fallback)")


def ptr(t):
    return C.c_void_p(0) if t is None else C.c_void_p(t.data_ptr())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def f32c(t, name="tensor"):
    """float32, contiguous, on device -- validated, not converted silently across devices."""
    require_device(t)
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()
