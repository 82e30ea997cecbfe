"""ctypes binding of libocrl_hip.so (include/ocrl_hip.h).  Fails loudly when the library is not built."""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_longlong, c_size_t, c_uint, c_ulonglong, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OCRL_HIP_LIB") or os.path.join(_HERE, "libocrl_hip.so")      # OCRL_HIP_LIB: development builds
_lib = None


class SlateConfig(ctypes.Structure):
    _fields_ = [(n, c_int) for n in ("obs_size", "obs_channels", "vocab_size", "d_model", "cnn_hidden", "num_slots",
                                      "num_iterations", "slot_size", "mlp_hidden", "num_dec_blocks", "num_dec_heads")] + \
               [("dropout", c_float), ("max_batch", c_int), ("use_bcdec", c_int), ("hard", c_int), ("num_slot_heads", c_int)]


class IodineConfig(ctypes.Structure):
    _fields_ = [(n, c_int) for n in ("obs_size", "obs_channels", "slot_size", "num_iterations", "num_slots")] + \
               [("sigma", c_float), ("beta", c_float), ("layer_norm", c_int), ("ref_mlp_hidden", c_int), ("max_batch", c_int)]


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C ocrl_amd/csrc).  ocrl_amd has no CPU fallback.")
    # PyTorch-ROCm carries its own copy of the HIP runtime; it has to be the one already in the process when this library resolves its
    # libamdhip64 dependency, or the two halves talk to two runtimes (observed: "no ROCm-capable device is detected" from the library
    # when it was loaded before torch)
    import torch  # noqa: F401
    L = ctypes.CDLL(LIB_PATH)
    p = c_void_p
    L.ocrl_last_error.restype = c_char_p
    L.ocrl_abi_version.restype = c_int
    L.ocrl_slate_config_size.restype = c_size_t
    if L.ocrl_slate_config_size() != ctypes.sizeof(SlateConfig):
        raise RuntimeError(f"libocrl_hip.so: ocrl_slate_config is {L.ocrl_slate_config_size()} bytes, this binding's SlateConfig {ctypes.sizeof(SlateConfig)}")
    L.ocrl_slate_create.argtypes = [POINTER(SlateConfig), POINTER(p)]
    L.ocrl_slate_destroy.argtypes = [p]
    L.ocrl_slate_destroy.restype = None
    L.ocrl_slate_param_count.argtypes = [p]
    L.ocrl_slate_param_info.argtypes = [p, c_int, c_char_p, c_int, POINTER(c_int * 4), POINTER(c_int), POINTER(c_longlong),
                                        POINTER(c_longlong), POINTER(c_int)]
    L.ocrl_slate_flat_size.argtypes = [p]
    L.ocrl_slate_flat_size.restype = c_longlong
    L.ocrl_slate_group_begin.argtypes = [p, c_int]
    L.ocrl_slate_group_begin.restype = c_longlong
    L.ocrl_slate_workspace_bytes.argtypes = [p]
    L.ocrl_slate_workspace_bytes.restype = c_size_t
    L.ocrl_slate_bind.argtypes = [p, p, p, p, p, p, c_size_t]
    L.ocrl_slate_forward.argtypes = [p, p, c_int, c_float, c_int, c_ulonglong, p, p, p, p]
    L.ocrl_slate_backward.argtypes = [p, p]
    L.ocrl_slate_generate.argtypes = [p, p]
    L.ocrl_slate_encode.argtypes = [p, p, c_int, c_ulonglong, p, p]
    L.ocrl_slate_encode_backward.argtypes = [p, p, p]
    L.ocrl_slate_freeze_weights.argtypes = [p, c_int]
    L.ocrl_slate_clip_adam.argtypes = [p, POINTER(c_float * 3), c_float, c_int, c_float, p]
    L.ocrl_slate_grad_norm.argtypes = [p, p]
    L.ocrl_slate_metrics.argtypes = [p]
    L.ocrl_slate_metrics.restype = p
    L.ocrl_slate_tensor.argtypes = [p, c_char_p, POINTER(p), POINTER(c_longlong)]
    L.ocrl_slate_dropout_mask.argtypes = [p, c_uint, c_longlong, p, p]
    L.ocrl_slate_soft_z.argtypes = [p, p]
    L.ocrl_gemm.argtypes = [p, p, p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, p, c_int, p, c_int, p, c_int,
                            c_int, p, p]
    L.ocrl_conv2d_fwd.argtypes = [p, p, p, p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, p, p]
    L.ocrl_conv2d_fwd_lowlat.argtypes = [p, p, p, p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, p, p]
    L.ocrl_conv2d_x3_ws_floats.restype = c_size_t
    L.ocrl_conv2d_fwd_x3.argtypes = [p, p, p, p, c_int, c_int, c_int, c_int, c_int, p, p]
    L.ocrl_conv2d_bwd_data_x3.argtypes = [p, p, p, p, c_int, c_int, c_int, c_int, p, p]
    L.ocrl_conv2d_bwd_weight_x3.argtypes = [p, p, p, c_int, c_int, c_int, c_int, p, c_size_t, p]
    L.ocrl_conv2d_bwd_data.argtypes = [p, p, p, p, c_int, c_int, c_int, c_int, p, p]
    L.ocrl_conv2d_wgrad_ws_floats.argtypes = [c_int, c_int, c_int, c_int, c_int]
    L.ocrl_conv2d_wgrad_ws_floats.restype = c_size_t
    L.ocrl_conv2d_bwd_weight.argtypes = [p, p, p, p, c_int, c_int, c_int, c_int, c_int, c_int, p, c_size_t, p]
    L.ocrl_layernorm_fwd.argtypes = [p, p, p, p, p, p, c_longlong, c_int, p]
    L.ocrl_layernorm_bwd.argtypes = [p, p, p, p, p, p, p, c_longlong, c_int, p, c_size_t, p]
    L.ocrl_attention_fwd.argtypes = [p, p, p, p, p, c_int, c_int, c_int, c_int, c_int, c_float, c_ulonglong, c_uint, p]
    L.ocrl_attention_bwd.argtypes = [p, p, p, p, p, p, p, p, p, p, c_int, c_int, c_int, c_int, c_int, c_float, c_ulonglong, c_uint, p]
    L.ocrl_obs_u8_to_f32.argtypes = [p, p, c_int, c_int, c_int, c_int, p]
    L.ocrl_prof_enable.argtypes = [c_uint]
    L.ocrl_prof_collect.argtypes = [POINTER(ctypes.c_double * 8), POINTER(c_longlong * 8), c_int]
    L.ocrl_iodine_create.argtypes = [POINTER(IodineConfig), POINTER(p)]
    L.ocrl_iodine_destroy.argtypes = [p]
    L.ocrl_iodine_destroy.restype = None
    L.ocrl_iodine_param_count.argtypes = [p]
    L.ocrl_iodine_param_info.argtypes = [p, c_int, c_char_p, c_int, POINTER(c_int * 4), POINTER(c_int), POINTER(c_longlong), POINTER(c_longlong)]
    L.ocrl_iodine_flat_size.argtypes = [p]
    L.ocrl_iodine_flat_size.restype = c_longlong
    L.ocrl_iodine_workspace_bytes.argtypes = [p]
    L.ocrl_iodine_workspace_bytes.restype = c_size_t
    L.ocrl_iodine_bind.argtypes = [p, p, p, p, p, p, c_size_t]
    L.ocrl_iodine_forward.argtypes = [p, p, c_int, c_ulonglong, p, p]
    L.ocrl_iodine_backward.argtypes = [p, p]
    L.ocrl_iodine_clip_adam.argtypes = [p, c_float, c_float, c_int, c_float, p]
    L.ocrl_iodine_grad_norm.argtypes = [p, p]
    L.ocrl_iodine_metrics.argtypes = [p]
    L.ocrl_iodine_metrics.restype = p
    L.ocrl_iodine_tensor.argtypes = [p, c_char_p, POINTER(p), POINTER(c_longlong)]
    L.ocrl_slot_attention_ws_floats.argtypes = [c_int, c_int, c_int, c_int, c_int]
    L.ocrl_slot_attention_ws_floats.restype = c_size_t
    L.ocrl_slot_attention_fwd.argtypes = [p, p, POINTER(p), p, p, c_int, c_int, c_int, c_int, c_int, c_int, p, c_size_t, p]
    L.ocrl_slot_attention_bwd.argtypes = [p, p, p, p, POINTER(p), c_int, c_int, c_int, c_int, c_int, c_int, p, c_size_t, p]
    L.ocrl_slot_attention_mh_ws_floats.argtypes = [c_int] * 7
    L.ocrl_slot_attention_mh_ws_floats.restype = c_size_t
    L.ocrl_slot_attention_mh_fwd.argtypes = [p, p, POINTER(p), p, p] + [c_int] * 7 + [p, c_size_t, p]
    L.ocrl_slot_attention_mh_bwd.argtypes = [p, p, p, p, POINTER(p)] + [c_int] * 7 + [p, c_size_t, p]
    L.ocrl_pool_transformer_ws_floats.argtypes = [c_int, c_int, c_int, c_int, c_int, c_int]
    L.ocrl_pool_transformer_ws_floats.restype = c_size_t
    L.ocrl_pool_transformer_fwd.argtypes = [p, POINTER(p), p, p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_ulonglong, p, c_size_t, p]
    L.ocrl_pool_transformer_bwd.argtypes = [p, p, POINTER(p), p, POINTER(p), c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_ulonglong, p,
                                            c_size_t, p]
    L.ocrl_pool_transformer_dropout_mask.argtypes = [c_int, c_int, c_longlong, c_float, c_ulonglong, p, p]
    L.ocrl_comm_unique_id.argtypes = [p, c_size_t]
    L.ocrl_comm_init.argtypes = [POINTER(p), c_int, c_int, p]
    L.ocrl_comm_allreduce.argtypes = [p, p, c_longlong, p]
    L.ocrl_comm_world.argtypes = [p]
    L.ocrl_comm_destroy.argtypes = [p]
    L.ocrl_comm_destroy.restype = None
    if L.ocrl_abi_version() != 5:
        raise RuntimeError("libocrl_hip.so ABI version mismatch")
    _lib = L
    return L


def check(rc):
    if rc:
        raise RuntimeError("ocrl_hip: " + lib().ocrl_last_error().decode())


def ptr(t):
    """device pointer of a torch tensor (or None)"""
    return None if t is None else ctypes.c_void_p(t.data_ptr())
