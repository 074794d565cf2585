"""ctypes binding of libqbnn_hip.so (C ABI: include/qbnn.h).

The product path has NO CPU fallback: if the library is missing or a call fails, this raises."""
import ctypes as C
import os

from . import build as _build

_LIB = None


class SampleParams(C.Structure):
    _fields_ = [("inv_noise_scale", C.c_float), ("mul_multiplier", C.c_float),
                ("z_sigma", C.c_int32), ("z_mul", C.c_int32),
                ("s_w", C.c_float), ("nzs_w", C.c_float),
                ("s_mul", C.c_float), ("nzs_mul", C.c_float),
                ("inv_s_add", C.c_float), ("z_add", C.c_int32),
                ("w_lo", C.c_int32), ("w_hi", C.c_int32)]


class ConvDesc(C.Structure):
    _fields_ = [("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32), ("Cout", C.c_int32),
                ("ksize", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
                ("s_x", C.c_float), ("z_x", C.c_int32),
                ("s_w", C.c_float), ("z_w", C.c_int32),
                ("s_y", C.c_float), ("z_y", C.c_int32),
                ("relu", C.c_int32), ("a_hi", C.c_int32), ("has_bias", C.c_int32), ("has_res", C.c_int32),
                ("s_r", C.c_float), ("z_r", C.c_int32),
                ("s_o", C.c_float), ("z_o", C.c_int32),
                ("x_is_centered_im2col", C.c_int32)]


class PostDesc(C.Structure):
    _fields_ = [("keep_prob", C.c_float), ("s_m", C.c_float), ("z_m", C.c_int32), ("drop_layer_id", C.c_uint32),
                ("add", C.c_int32), ("s_a", C.c_float), ("s_b", C.c_float), ("z_b", C.c_int32), ("s_o", C.c_float), ("z_o", C.c_int32)]


class DropoutDesc(C.Structure):
    _fields_ = [("keep_prob", C.c_float), ("s_m", C.c_float), ("z_m", C.c_int32), ("layer_id", C.c_uint32)]


class DropDesc(C.Structure):
    """qbnn_drop_desc: a BernoulliDropout behind a conv of a fused block (conv_resnet_mc)."""
    _fields_ = [("keep_prob", C.c_float), ("s_m", C.c_float), ("z_m", C.c_int32), ("s_out", C.c_float), ("layer_id", C.c_uint32),
                ("mask_in", C.c_void_p)]


class SamplerLayer(C.Structure):
    _fields_ = [("mu_packed", C.c_void_p), ("sigma_packed", C.c_void_p), ("w_out", C.c_void_p), ("w_sample_stride", C.c_int64),
                ("cout", C.c_int32), ("k", C.c_int32), ("krow", C.c_int32), ("layout", C.c_int32),
                ("layer_id", C.c_uint32), ("params", SampleParams)]


class QatWLayer(C.Structure):
    """qbnn_qat_wlayer (include/qbnn.h): one stochastic layer of qbnn_qat_weights_mc."""
    _fields_ = [("mu", C.c_void_p), ("sg", C.c_void_p), ("st_w", C.c_void_p), ("st_s", C.c_void_p), ("st_m", C.c_void_p), ("st_a", C.c_void_p),
                ("cmm", C.c_float * 4), ("n", C.c_int32), ("Cout", C.c_int32), ("Cin", C.c_int32), ("KS", C.c_int32), ("layer_id", C.c_uint32),
                ("qmin", C.c_int32), ("qmax", C.c_int32), ("blk0", C.c_int32), ("nblk", C.c_int32), ("pm", C.c_void_p), ("pa", C.c_void_p),
                ("W", C.c_void_p), ("q8", C.c_void_p), ("scale", C.c_void_p), ("zp", C.c_void_p)]


class F32WLayer(C.Structure):
    """qbnn_f32_wlayer (include/qbnn.h): one layer of qbnn_sample_weights_f32_batch."""
    _fields_ = [("mu", C.c_void_p), ("sigma", C.c_void_p), ("w", C.c_void_p), ("n", C.c_int32), ("Cout", C.c_int32), ("Cin", C.c_int32), ("KS", C.c_int32),
                ("layer_id", C.c_uint32), ("blk0", C.c_int32), ("nblk", C.c_int32)]


class BlockDesc(C.Structure):
    _fields_ = [("w_a", C.c_void_p), ("w_a_sample_stride", C.c_int64), ("bias_a", C.c_void_p),
                ("s_wa", C.c_float), ("z_wa", C.c_int32), ("s_a", C.c_float), ("z_a", C.c_int32),
                ("w_b", C.c_void_p), ("w_b_sample_stride", C.c_int64), ("bias_b", C.c_void_p),
                ("s_wb", C.c_float), ("z_wb", C.c_int32), ("s_b", C.c_float), ("z_b", C.c_int32),
                ("s_o", C.c_float), ("z_o", C.c_int32), ("w_layout", C.c_int32), ("flags", C.c_int32)]


class MlpLayer(C.Structure):
    _fields_ = [("mu", C.c_void_p), ("sigma", C.c_void_p), ("bias", C.c_void_p), ("out_features", C.c_int32), ("in_features", C.c_int32),
                ("layer_id", C.c_uint32)]


class DownDesc(C.Structure):
    _fields_ = [("blk", BlockDesc), ("w_s", C.c_void_p), ("w_s_sample_stride", C.c_int64), ("bias_s", C.c_void_p),
                ("s_ws", C.c_float), ("z_ws", C.c_int32), ("s_s", C.c_float), ("z_s", C.c_int32)]


class HeadDesc(C.Structure):
    _fields_ = [("B", C.c_int32), ("k", C.c_int32), ("C", C.c_int32), ("N", C.c_int32),
                ("s_x", C.c_float), ("z_x", C.c_int32),
                ("s_w", C.c_float), ("z_w", C.c_int32),
                ("s_y", C.c_float), ("z_y", C.c_int32),
                ("a_hi", C.c_int32), ("has_bias", C.c_int32)]


class ChainCall(C.Structure):
    _fields_ = [("x", C.c_void_p), ("x_sample_stride", C.c_int64), ("s_x", C.c_float), ("z_x", C.c_int32),
                ("blocks", C.POINTER(BlockDesc)), ("y", C.c_void_p), ("y_sample_stride", C.c_int64), ("n_samples", C.c_int32),
                ("im2col", C.c_void_p), ("w0_packed", C.c_void_p), ("w0_sample_stride", C.c_int64), ("bias0", C.c_void_p),
                ("s_in", C.c_float), ("s_w0", C.c_float), ("z_w0", C.c_int32), ("s_y0", C.c_float), ("z_y0", C.c_int32)]


class DownCall(C.Structure):
    _fields_ = [("x", C.c_void_p), ("x_sample_stride", C.c_int64), ("s_x", C.c_float), ("z_x", C.c_int32),
                ("desc", C.POINTER(DownDesc)), ("y", C.c_void_p), ("y_sample_stride", C.c_int64), ("n_samples", C.c_int32)]


class HeadCall(C.Structure):
    _fields_ = [("x", C.c_void_p), ("x_sample_stride", C.c_int64), ("w", C.c_void_p), ("w_sample_stride", C.c_int64), ("bias", C.c_void_p),
                ("probs", C.c_void_p), ("n_samples", C.c_int32), ("desc", C.POINTER(HeadDesc))]


EXPORTS = ["qbnn_mlp_bbb_f32_mc", "qbnn_mlp_bbb_f32_workspace_floats", "qbnn_packed_weight_bytes", "qbnn_pack_weights_host", "qbnn_sample_weights_i8", "qbnn_sample_weights_i8_multi", "qbnn_set_device_noise_source", "qbnn_conv2d_i8_mc", "qbnn_conv2d_i8_post_mc", "qbnn_conv_pool_drop_i8_mc", "qbnn_linear_i8_mc", "qbnn_im2col5x5_c1", "qbnn_conv_c1_pool_i8_mc",
           "qbnn_block_chain_i8_mc", "qbnn_stem_chain_i8_mc", "qbnn_block_down_i8_mc", "qbnn_block_chain_drop_i8_mc", "qbnn_stem_chain_drop_i8_mc", "qbnn_block_down_drop_i8_mc", "qbnn_block_chain_i8_multi", "qbnn_block_down_i8_multi", "qbnn_chain_multi_args_bytes", "qbnn_down_multi_args_bytes", "qbnn_block_chain_i8_multi_prepare", "qbnn_block_chain_i8_multi_launch", "qbnn_block_down_i8_multi_prepare", "qbnn_block_down_i8_multi_launch", "qbnn_head_i8_multi", "qbnn_quantize_im2col3x3_c3_multi",
           "qbnn_quantize_input_nchw", "qbnn_im2col3x3_c3", "qbnn_conv2d_i8_generic_mc", "qbnn_conv2d_i8_generic_scalar_mc", "qbnn_dropout_q_mc", "qbnn_maxpool2_q_mc",
           "qbnn_dequant_softmax_mc", "qbnn_flatten_nchw_mc", "qbnn_flatten_nchw_rows_mc", "qbnn_add_relu_q_mc", "qbnn_sample_weights_f32", "qbnn_linear_f32_mc", "qbnn_head_i8_mc", "qbnn_reduce_moments", "qbnn_finalize_moments", "qbnn_classification_metrics", "qbnn_regression_metrics",
           "qbnn_conv2d_f32_mc", "qbnn_conv2d_f32_fused_mc", "qbnn_conv2d_f32_blocks", "qbnn_observe_partials_f32_mc", "qbnn_grid_to_i8_mc", "qbnn_fake_quant_ex_f32_mc", "qbnn_conv2d_q8_blocks", "qbnn_conv2d_q8_f32_mc", "qbnn_add_q8_blocks", "qbnn_add_q8_f32_mc", "qbnn_fake_quant_add_q8_mc", "qbnn_qat_weights_mc", "qbnn_sample_weights_f32_batch", "qbnn_affine_f32_mc", "qbnn_pool2d_f32_mc", "qbnn_flatten_nchw_f32_mc", "qbnn_softmax_f32_mc", "qbnn_observe_f32_mc", "qbnn_fake_quant_f32_mc", "qbnn_sample_weights_f32_strided", "qbnn_sample_weights_f32_ohwi",
           "qbnn_dropout_mask_f32_mc", "qbnn_dropout_f32_mc", "qbnn_conv2d_f32_drop_mc",
           "qbnn_last_error", "qbnn_version"]


ABI_VERSION = 2      # = QBNN_ABI_VERSION of include/qbnn.h


def lib_path():
    return _build.LIB


def lib():
    global _LIB
    if _LIB is None:
        path = _build.LIB
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(there is no CPU fallback for the qbnn HIP path)")
        L = C.CDLL(path)
        L.qbnn_last_error.restype = C.c_char_p
        if L.qbnn_version() != ABI_VERSION:      # a stale in-tree build against an older include/qbnn.h: argument lists differ
            raise RuntimeError(f"{path} reports ABI version {L.qbnn_version()}, this binding is written against {ABI_VERSION} "
                               "(include/qbnn.h: QBNN_ABI_VERSION): rebuild with `python -c 'import __graft_entry__ as g; g.build()'`")
        L.qbnn_packed_weight_bytes.restype = C.c_size_t
        L.qbnn_packed_weight_bytes.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32]
        vp, i32, i64, u32, u64, f = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32, C.c_uint64, C.c_float
        L.qbnn_pack_weights_host.argtypes = [vp, i32, i32, i32, i32, vp]
        L.qbnn_sample_weights_i8.argtypes = [vp, vp, i32, i32, i32, i32, C.POINTER(SampleParams), u64, u32, u32, i32, vp, vp, i64, vp]
        L.qbnn_set_device_noise_source.argtypes = [vp]
        L.qbnn_sample_weights_i8_multi.argtypes = [C.POINTER(SamplerLayer), i32, u64, u32, i32, vp]
        L.qbnn_conv2d_i8_mc.argtypes = [vp, i64, vp, i64, vp, vp, i64, vp, i64, i32, C.POINTER(ConvDesc), vp]
        L.qbnn_conv2d_i8_post_mc.argtypes = [vp, i64, vp, i64, vp, vp, i64, i32, C.POINTER(ConvDesc), C.POINTER(PostDesc), vp, vp, i64, u64, u32, vp]
        L.qbnn_conv_pool_drop_i8_mc.argtypes = [vp, i64, vp, i64, vp, vp, i64, i32, i32, C.POINTER(ConvDesc), i32, C.POINTER(DropoutDesc), vp,
                                                C.POINTER(DropoutDesc), vp, f, i32, u64, u32, vp]
        L.qbnn_im2col5x5_c1.argtypes = [vp, i32, i32, i32, i32, vp, vp]
        L.qbnn_conv_c1_pool_i8_mc.argtypes = [vp, i64, vp, i64, vp, vp, i64, i32, C.POINTER(ConvDesc), vp]
        L.qbnn_linear_i8_mc.argtypes = [vp, i64, i32, vp, i64, vp, vp, i64, i32, i32, C.POINTER(ConvDesc), C.POINTER(DropoutDesc), vp, u64, u32, vp]
        L.qbnn_block_chain_i8_mc.argtypes = [vp, i64, f, i32, i32, i32, i32, i32, C.POINTER(BlockDesc), i32, vp, i64, i32, vp]
        L.qbnn_stem_chain_i8_mc.argtypes = [vp, i32, vp, i64, vp, f, f, i32, f, i32, i32, C.POINTER(BlockDesc), i32, vp, i64, i32, vp]
        L.qbnn_block_down_i8_mc.argtypes = [vp, i64, f, i32, i32, i32, i32, i32, C.POINTER(DownDesc), vp, i64, i32, vp]
        L.qbnn_block_chain_drop_i8_mc.argtypes = [vp, i64, f, i32, i32, i32, i32, i32, C.POINTER(BlockDesc), C.POINTER(DropDesc), i32, vp, i64, i32, C.c_uint64, C.c_uint32, vp]
        L.qbnn_stem_chain_drop_i8_mc.argtypes = [vp, i32, vp, i64, vp, f, f, i32, f, i32, i32, C.POINTER(DropDesc), C.POINTER(BlockDesc), C.POINTER(DropDesc), i32, vp, i64, i32,
                                                 C.c_uint64, C.c_uint32, vp]
        L.qbnn_block_down_drop_i8_mc.argtypes = [vp, i64, f, i32, i32, i32, i32, i32, C.POINTER(DownDesc), C.POINTER(DropDesc), vp, i64, i32, C.c_uint64, C.c_uint32, vp]
        L.qbnn_block_chain_i8_multi.argtypes = [C.POINTER(ChainCall), i32, i32, i32, i32, i32, i32, i32, vp]
        L.qbnn_block_down_i8_multi.argtypes = [C.POINTER(DownCall), i32, i32, i32, i32, i32, vp]
        L.qbnn_chain_multi_args_bytes.restype = C.c_size_t
        L.qbnn_chain_multi_args_bytes.argtypes = [i32, i32]
        L.qbnn_down_multi_args_bytes.restype = C.c_size_t
        L.qbnn_down_multi_args_bytes.argtypes = [i32]
        L.qbnn_block_chain_i8_multi_prepare.argtypes = [C.POINTER(ChainCall), i32, i32, i32, i32, i32, vp, vp]
        L.qbnn_block_chain_i8_multi_launch.argtypes = [vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp]
        L.qbnn_block_down_i8_multi_prepare.argtypes = [C.POINTER(DownCall), i32, i32, i32, vp, vp]
        L.qbnn_block_down_i8_multi_launch.argtypes = [vp, i32, i32, i32, i32, i32, i32, vp]
        L.qbnn_head_i8_multi.argtypes = [C.POINTER(HeadCall), i32, vp]
        L.qbnn_quantize_im2col3x3_c3_multi.argtypes = [vp, i32, i32, i32, C.POINTER(f), C.POINTER(i32), i32, i32, vp, i64, vp]
        L.qbnn_conv2d_i8_generic_mc.argtypes = [vp, i64, vp, i64, vp, vp, i64, i32, C.POINTER(ConvDesc), vp]
        L.qbnn_conv2d_i8_generic_scalar_mc.argtypes = L.qbnn_conv2d_i8_generic_mc.argtypes
        L.qbnn_dropout_q_mc.argtypes = [vp, i64, i32, i32, i32, f, f, i32, f, i32, i32, u64, u32, u32, vp, vp, i64, i32, vp]
        L.qbnn_maxpool2_q_mc.argtypes = [vp, i64, i32, i32, i32, i32, i32, vp, i64, i32, vp]
        L.qbnn_add_relu_q_mc.argtypes = [vp, i64, f, i32, vp, i64, f, i32, vp, i64, i64, f, i32, i32, i32, i32, vp]
        L.qbnn_dequant_softmax_mc.argtypes = [vp, i64, i32, i32, f, i32, vp, i32, vp]
        L.qbnn_sample_weights_f32.argtypes = [vp, vp, i64, u64, u32, u32, i32, vp, vp, vp]
        L.qbnn_linear_f32_mc.argtypes = [vp, i64, vp, i64, vp, vp, i64, i32, i32, i32, i32, i32, vp]
        L.qbnn_classification_metrics.argtypes = [vp, vp, i32, i32, vp, vp]
        L.qbnn_regression_metrics.argtypes = [vp, vp, vp, i32, vp, vp]
        L.qbnn_flatten_nchw_mc.argtypes = [vp, i64, i32, i32, i32, vp, i64, i32, vp]
        L.qbnn_flatten_nchw_rows_mc.argtypes = [vp, i64, i32, i32, i32, i32, vp, i64, i32, i32, vp]
        L.qbnn_quantize_input_nchw.argtypes = [vp, i32, i32, i32, i32, f, i32, i32, vp, vp]
        L.qbnn_im2col3x3_c3.argtypes = [vp, i32, i32, i32, i32, vp, vp]
        L.qbnn_head_i8_mc.argtypes = [vp, i64, vp, i64, vp, vp, i32, C.POINTER(HeadDesc), vp]
        L.qbnn_reduce_moments.argtypes = [vp, i32, i64, i32, vp, i32, vp, vp, vp]
        L.qbnn_finalize_moments.argtypes = [vp, i64, i32, vp, vp, vp]
        L.qbnn_conv2d_f32_fused_mc.argtypes = [vp, i64, vp, i64, vp, vp, vp, vp, vp, i64, vp, i64, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp]
        L.qbnn_conv2d_f32_blocks.argtypes = [i32, i32, i32, i32, i32, i32, i32]
        L.qbnn_conv2d_f32_blocks.restype = i32
        L.qbnn_observe_partials_f32_mc.argtypes = [vp, i32, i32, vp, f, i32, i32, vp, vp, vp]
        L.qbnn_grid_to_i8_mc.argtypes = [vp, i64, i64, vp, vp, vp, i32, vp]
        L.qbnn_fake_quant_ex_f32_mc.argtypes = [vp, i64, vp, i64, i64, vp, vp, i32, i32, i32, vp, i32, vp]
        L.qbnn_conv2d_q8_blocks.argtypes = [i32, i32, i32, i32, i32, i32, i32, i32]
        L.qbnn_conv2d_q8_blocks.restype = i32
        L.qbnn_qat_weights_mc.argtypes = [vp, i32, i32, C.c_float, u64, u32, i32, vp]
        L.qbnn_sample_weights_f32_batch.argtypes = [vp, i32, i32, u64, u32, i32, vp]
        L.qbnn_add_q8_blocks.argtypes = [i64]
        L.qbnn_add_q8_blocks.restype = i32
        L.qbnn_add_q8_f32_mc.argtypes = [vp, i64, vp, vp, i64, vp, vp, i64, i64, i32, vp, vp]
        L.qbnn_fake_quant_add_q8_mc.argtypes = [vp, i64, vp, vp, i64, vp, vp, i64, i64, vp, vp, i32, i32, i32, vp, i32, vp]
        L.qbnn_conv2d_q8_f32_mc.argtypes = [vp, i64, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, i64] + [i32] * 10 + [vp, vp]
        L.qbnn_conv2d_f32_mc.argtypes = [vp, i64, vp, i64, vp, vp, i64, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp]
        L.qbnn_affine_f32_mc.argtypes = [vp, i64, vp, i64, vp, vp, vp, i64, i64, i32, i32, i32, i32, vp]
        L.qbnn_pool2d_f32_mc.argtypes = [vp, i64, vp, i64, i32, i32, i32, i32, i32, i32, i32, vp]
        L.qbnn_flatten_nchw_f32_mc.argtypes = [vp, i64, i32, i32, i32, vp, i64, i32, vp]
        L.qbnn_softmax_f32_mc.argtypes = [vp, i64, i32, i32, vp, i32, vp]
        L.qbnn_observe_f32_mc.argtypes = [vp, i64, i64, i32, vp, f, i32, i32, vp, vp, vp, vp]
        L.qbnn_fake_quant_f32_mc.argtypes = [vp, i64, vp, i64, i64, vp, vp, i32, i32, i32, i32, vp]
        L.qbnn_sample_weights_f32_strided.argtypes = [vp, i64, vp, i64, i64, u64, u32, u32, i32, vp, vp, vp]
        L.qbnn_mlp_bbb_f32_mc.argtypes = [vp, i32, C.POINTER(MlpLayer), u64, u32, i32, vp, vp, vp, vp]
        L.qbnn_mlp_bbb_f32_workspace_floats.argtypes = [C.POINTER(MlpLayer)]
        L.qbnn_mlp_bbb_f32_workspace_floats.restype = i64
        L.qbnn_sample_weights_f32_ohwi.argtypes = [vp, i64, vp, i64, i32, i32, i32, u64, u32, u32, i32, vp, vp, vp]
        L.qbnn_dropout_mask_f32_mc.argtypes = [i64, f, u64, u32, u32, i32, vp, vp]
        L.qbnn_dropout_f32_mc.argtypes = [vp, i64, vp, i32, i32, i32, f, vp, i64, i32, vp, i64, i32, vp]
        L.qbnn_conv2d_f32_drop_mc.argtypes = [vp, i64, vp, i64, vp, vp, vp, vp, f, vp, i64, vp, i64, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp]
        _LIB = L
    return _LIB


def check(rc):
    if rc != 0:
        raise RuntimeError(f"qbnn error {rc}: {lib().qbnn_last_error().decode()}")


def ptr(t):
    """Device (or host) address of a torch tensor / None."""
    return None if t is None else C.c_void_p(t.data_ptr())


def current_stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
