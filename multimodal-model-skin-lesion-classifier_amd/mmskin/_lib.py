"""ctypes binding of libmmskin_hip.so (the C ABI declared in include/mmskin.h).

There is no CPU fallback: if the shared library is missing or a call fails, the
caller gets an exception.  PyTorch is used only for device memory, streams and
autograd plumbing around these calls.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "libmmskin_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(os.path.dirname(_HERE)), "include", "mmskin.h")

F32, BF16 = 0, 1

_lib = None

c_f = ctypes.c_void_p          # device pointers travel as void*
_i, _i64, _f = ctypes.c_int, ctypes.c_int64, ctypes.c_float
_u64 = ctypes.c_uint64
_P = ctypes.c_void_p

_SIGNATURES = {
    "mmskin_last_error": (ctypes.c_char_p, []),
    "mmskin_version": (_i, []),
    "mmskin_backbone_create": (_i, [ctypes.c_char_p, _i, _i, _i, _i, ctypes.POINTER(_P)]),
    "mmskin_backbone_destroy": (None, [_P]),
    "mmskin_backbone_num_tensors": (_i, [_P, _i]),
    "mmskin_backbone_tensor_info": (_i, [_P, _i, _i, ctypes.c_char_p, _i, ctypes.POINTER(_i64),
                                         ctypes.POINTER(_i64), ctypes.POINTER(_i), ctypes.POINTER(_i64)]),
    "mmskin_backbone_param_numel": (_i64, [_P]),
    "mmskin_backbone_buffer_numel": (_i64, [_P]),
    "mmskin_backbone_workspace_bytes": (_i64, [_P]),
    "mmskin_backbone_feature_dim": (_i, [_P]),
    "mmskin_backbone_feature_hw": (_i, [_P, _P, _P]),
    "mmskin_backbone_set_option": (_i, [_P, ctypes.c_char_p, _i]),
    "mmskin_backbone_set_pointer": (_i, [_P, ctypes.c_char_p, _P]),
    "mmskin_backbone_num_grad_segments": (_i, [_P, ctypes.POINTER(ctypes.c_int)]),
    "mmskin_backbone_grad_segment": (_i, [_P, _i, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]),
    "mmskin_backbone_wait_grad_segment": (_i, [_P, _i, _P]),
    "mmskin_backbone_last_conv_shape": (_i, [_P, _P, _P, _P]),
    "mmskin_backbone_last_conv_export": (_i, [_P, _P, _P, _P]),
    "mmskin_backbone_last_conv_grad": (_i, [_P, _P, _P, _P, _P]),
    "mmskin_mdnet_fuse_forward": (_i, [_P] * 5 + [_i64, _i, _P]),
    "mmskin_mdnet_fuse_backward": (_i, [_P] * 9 + [_i64, _i, _P]),
    "mmskin_backbone_profile_enable": (_i, [_P, _i]),
    "mmskin_backbone_profile_read": (_i, [_P, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                                          ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_i64)]),
    "mmskin_backbone_num_units": (_i, [_P]),
    "mmskin_backbone_unit_info": (_i, [_P, _i, ctypes.c_char_p, _i, ctypes.POINTER(_i64)]),
    "mmskin_backbone_forward": (_i, [_P, _P, _P, _P, _P, _P, _i, _P]),
    "mmskin_backbone_forward_u8": (_i, [_P, _P, _P, _P, _P, _P, _P, _i, _P]),
    "mmskin_backbone_backward": (_i, [_P, _P, _P, _P, _P, _P]),
    "mmskin_conv2d_workspace_bytes": (_i64, [_i] * 9),
    "mmskin_conv2d_forward": (_i, [_P, _P, _P] + [_i] * 10 + [_P, _P]),
    "mmskin_conv2d_time": (ctypes.c_double, [_i] * 11 + [_P, _P]),
    "mmskin_conv2d_wgrad_time": (ctypes.c_double, [_i] * 11 + [_P, _P]),
    "mmskin_conv2d_dgrad_time": (ctypes.c_double, [_i] * 11 + [_P, _P]),
    "mmskin_conv2d_backward": (_i, [_P] * 5 + [_i] * 10 + [_P, _P]),
    "mmskin_conv_pipe_launches": (_i64, []),
    "mmskin_wgrad_ring_launches": (_i64, []),
    "mmskin_conv3x3_c64_launches": (_i64, []),
    "mmskin_stem7x7_launches": (_i64, []),
    "mmskin_conv2d_dgrad_fused_rows": (_i, [_i] * 9),
    "mmskin_conv2d_dgrad_fused": (_i, [_P] * 8 + [_i] * 9 + [_P, _P]),
    "mmskin_wgrad3_ring_launches": (_i64, []),
    "mmskin_abn_workspace_bytes": (_i64, [_i] * 5),
    "mmskin_abn_backward": (_i, [_P] * 6 + [_i] * 5 + [_P, _P, _P, _P]),
    "mmskin_adam_step": (_i, [_P] * 4 + [_i64] + [ctypes.c_double] * 5 + [_i64, _P]),
    "mmskin_abn_backward_kept_gram": (_i, [_P] * 6 + [_i] * 5 + [_P, _P, _P, _P]),
    "mmskin_conv1x1_gram_stats": (_i, [_P, _P] + [_i] * 5 + [_P, _P, _P, _P]),
    "mmskin_batchnorm_workspace_bytes": (_i64, [_i] * 4),
    "mmskin_batchnorm_forward": (_i, [_P] * 8 + [_i] * 4 + [_f, _f, _i, _i, _P, _P]),
    "mmskin_batchnorm_backward": (_i, [_P] * 9 + [_i] * 6 + [_P, _P]),
    "mmskin_stem_workspace_bytes": (_i64, [_i] * 3),
    "mmskin_stem_forward": (_i, [_P] * 5 + [_i] * 3 + [_f, _i, _P, _P]),
    "mmskin_stem_backward": (_i, [_P] * 8 + [_i] * 3 + [_f, _i, _P, _P]),
    "mmskin_linear_forward": (_i, [_P] * 4 + [_i] * 4 + [_P]),
    "mmskin_linear_backward": (_i, [_P] * 8 + [_i] * 3 + [_P]),
    "mmskin_linear_gelu_backward": (_i, [_P] * 8 + [_i] * 3 + [_P]),
    "mmskin_linear_x16_pitch": (_i, [_i, _i, _i]),
    "mmskin_linear_forward_keep": (_i, [_P] * 6 + [_i] * 4 + [_P]),
    "mmskin_linear_backward_keep": (_i, [_P] * 9 + [_i] * 3 + [_P]),
    "mmskin_gelu_forward_bf16": (_i, [_P, _P, _i64, _i, _i, _P]),
    "mmskin_linear_forward_x16": (_i, [_P] * 5 + [_i] * 4 + [_P]),
    "mmskin_layernorm_forward": (_i, [_P] * 6 + [_i, _i, _f, _i, _P]),
    "mmskin_layernorm_backward": (_i, [_P] * 9 + [_i, _i, _i, _P]),
    "mmskin_sigmoid_gate_forward": (_i, [_P] * 3 + [_i64, _P]),
    "mmskin_sigmoid_gate_backward": (_i, [_P] * 5 + [_i64, _P]),
    "mmskin_gated_mix_forward": (_i, [_P] * 4 + [_i64, _P]),
    "mmskin_gated_mix_backward": (_i, [_P] * 7 + [_i64, _P]),
    "mmskin_metablock_gate_forward": (_i, [_P] * 4 + [_i64, _P]),
    "mmskin_metablock_gate_backward": (_i, [_P] * 7 + [_i64, _P]),
    "mmskin_dropout_forward": (_i, [_P] * 3 + [_i64, _f, _u64, _u64, _P]),
    "mmskin_attn_dropout_forward": (_i, [_P] * 3 + [_i64, _i, _f, _u64, _u64, _P]),
    "mmskin_dropout_backward": (_i, [_P] * 3 + [_i64, _f, _P]),
    "mmskin_concat2_forward": (_i, [_P] * 3 + [_i] * 3 + [_P]),
    "mmskin_concat2_backward": (_i, [_P] * 3 + [_i] * 3 + [_P]),
    "mmskin_attention_forward": (_i, [_P] * 5 + [_i] * 4 + [_f, _u64, _u64, _P]),
    "mmskin_attention_backward": (_i, [_P] * 8 + [_i] * 4 + [_f, _u64, _u64, _P]),
    "mmskin_bmm": (_i, [_P] * 3 + [_i] * 4 + [_i64] * 8 + [_P]),
    "mmskin_softmax_forward": (_i, [_P] * 4 + [_i64, _i, _i64, _f, _i, _P]),
    "mmskin_gelu_tanh_forward": (_i, [_P, _P, _i64, _P]),
    "mmskin_gelu_tanh_backward": (_i, [_P, _P, _P, _i64, _P]),
    "mmskin_colsum": (_i, [_P, _P, _i, _i, _P]),
    "mmskin_dwconv3_scratch_floats": (_i64, [_i, _i, _i, _i]),
    "mmskin_dwconv3_forward": (_i, [_P] * 4 + [_i] * 4 + [_P]),
    "mmskin_dwconv3_backward": (_i, [_P] * 7 + [_i] * 4 + [_P]),
    "mmskin_conv_pos_enc_forward": (_i, [_P] * 5 + [_i] * 4 + [_P]),
    "mmskin_conv_pos_enc_backward": (_i, [_P] * 8 + [_i] * 4 + [_P]),
    "mmskin_scale_add_forward": (_i, [_P] * 4 + [_i64, _i, _P]),
    "mmskin_scale_mul": (_i, [_P] * 3 + [_i64, _i, _i, _P]),
    "mmskin_token_mean_forward": (_i, [_P, _P, _i, _i, _i, _i, _P]),
    "mmskin_token_mean_backward": (_i, [_P, _P, _i, _i, _i, _i, _P]),
    "mmskin_softmax_backward": (_i, [_P] * 3 + [_i64, _i, _f, _P]),
    "mmskin_add": (_i, [_P, _P, _P, _i64, _i64, _P]),
    "mmskin_gelu_forward": (_i, [_P, _P, _i64, _P]),
    "mmskin_gelu_backward": (_i, [_P, _P, _P, _i64, _P]),
    "mmskin_embedding_forward": (_i, [_P] * 3 + [_i] * 4 + [_P]),
    "mmskin_embedding_backward": (_i, [_P] * 3 + [_i] * 4 + [_P]),
    "mmskin_direct_conv2d_forward": (_i, [_P] * 4 + [_i] * 10 + [_P]),
    "mmskin_direct_conv2d_backward": (_i, [_P] * 5 + [_i] * 9 + [_P]),
    "mmskin_linear_forward_ex": (_i, [_P, _i, _P, _P, _P, _i, _i, _i, _i, _i, _P]),
    "mmskin_attention_rows_forward": (_i, [_P] * 5 + [_i] * 4 + [_P, _P, _f, _f, _u64, _u64, _P]),
    "mmskin_attention_rows_backward": (_i, [_P] * 9 + [_i] * 4 + [_P, _P, _f, _f, _u64, _u64, _P]),
    "mmskin_window_attention_forward": (_i, [_P] * 5 + [_i] * 6 + [_i64] * 4 + [_f, _f, _u64, _u64, _P]),
    "mmskin_window_attention_backward": (_i, [_P] * 9 + [_i] * 6 + [_i64] * 4 + [_f, _f, _u64, _u64, _P]),
    "mmskin_channel_attention_scratch_floats": (_i64, [_i, _i, _i]),
    "mmskin_channel_attention_forward": (_i, [_P] * 6 + [_i] * 4 + [_i64] * 4 + [_f, _P]),
    "mmskin_channel_attention_backward": (_i, [_P] * 9 + [_i] * 4 + [_i64] * 4 + [_f, _P]),
    "mmskin_linear_lane": (_i, [_P, _i, _P, _i, _P, _P, _P, _f, _u64, _u64, _P, _i, _i, _i, _i, _i, _P]),
    "mmskin_layernorm_forward_mixed": (_i, [_P] * 5 + [_i, _i, _f, _P]),
    "mmskin_flash_attention_forward": (_i, [_P] * 7 + [_i] * 4 + [_P, _i, _f, _i, _f, _u64, _u64, _P]),
    "mmskin_flash_attention_backward": (_i, [_P] * 14 + [_i] * 4 + [_P, _f, _i, _f, _u64, _u64, _P]),
    "mmskin_set_linear_dtype": (_i, [_i]),
    "mmskin_get_linear_dtype": (_i, []),
    "mmskin_im2col_forward": (_i, [_P] + [_i] * 8 + [_P, _P]),
    "mmskin_im2col_backward": (_i, [_P] + [_i] * 8 + [_P, _P]),
    "mmskin_resize_u8": (_i, [_P, _i, _i, _i, _P, _i, _i, _P]),
    "mmskin_metadata_encode": (_i, [_P, _i, _P, _i, _P, _i, _P, _P, _f, _P, _i, _P]),
    "mmskin_pool_gap_forward": (_i, [_P] * 3 + [_i] * 5 + [_P]),
    "mmskin_pool_gap_backward": (_i, [_P] * 3 + [_i] * 5 + [_P]),
}


class MMSkinError(RuntimeError):
    pass


def declared_symbols():
    """Every function name include/mmskin.h declares (used by the export test)."""
    with open(HEADER_PATH) as f:
        text = f.read()
    return sorted(set(re.findall(r"\b(mmskin_[a-z0-9_]+)\s*\(", text)))


def load():
    """Load the shared library (once) and attach prototypes.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MMSkinError(
            f"{LIB_PATH} not found: build it with `python __graft_entry__.py` (or `make -C "
            f"{os.path.join(os.path.dirname(_HERE), 'csrc')}`).  There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().mmskin_last_error()
        raise MMSkinError(f"mmskin error {rc}: {msg.decode() if msg else ''}")


def ptr(t):
    """Device pointer of a tensor (or None)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


_raw_stream = None


def stream():
    """hipStream_t of torch's current stream on the current device (raw handle: torch.cuda.current_stream() builds a Stream object
    and resolves the device index in Python, ~8 us per op on a launch-bound step)"""
    global _raw_stream
    if _raw_stream is None:
        import torch
        get, dev = getattr(torch._C, "_cuda_getCurrentRawStream", None), getattr(torch._C, "_cuda_getDevice", None)
        if get is not None and dev is not None:
            _raw_stream = lambda: get(dev())
        else:
            _raw_stream = lambda: torch.cuda.current_stream().cuda_stream
    return ctypes.c_void_p(_raw_stream())


_fns = {}


def call(name, *args):
    fn = _fns.get(name)
    if fn is None:
        fn = _fns[name] = getattr(load(), name)
    rc = fn(*args)
    if rc != 0:
        check(rc)
