"""
ctypes binding of liblidbox_hip.so (include/lidbox_hip.h).

The library is REQUIRED: there is no CPU or eager-PyTorch fallback anywhere in lidbox_amd.
If the .so is missing or does not load, importing this module raises.
"""
import ctypes as C
import os

# torch first: it ships its own libamdhip64.so.7; loading ours before it would put a second HIP
# runtime in the process (and hipGetDevice in that copy then reports "no ROCm-capable device").
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# LIDBOX_HIP_LIB: debugging aid only (instrumented builds from tools/ab_build.py); the product loads the in-tree library
LIB_PATH = os.environ.get("LIDBOX_HIP_LIB") or os.path.join(_HERE, "csrc", "liblidbox_hip.so")

ABI_VERSION = 1

FEAT_SPECTROGRAM, FEAT_MEL, FEAT_LOGMEL, FEAT_MFCC = 0, 1, 2, 3
SRC_F32, SRC_PCM16 = 0, 1
EPI_NONE, EPI_BIAS, EPI_BIAS_RELU, EPI_RELU_MASK, EPI_ACCUM, EPI_ACCUM_RELU_MASK, EPI_ACCUM_RELU, EPI_RELU = range(8)
EPI_MASK_BF16 = 0x100            # flag for lidbox_gemm_bf16s_nt: the ReLU-mask source is bfloat16 data


class LidboxHipError(RuntimeError):
    pass


class Rows(C.Structure):
    """lidbox_rows_t / lidbox_rows_out_t (identical layout)."""
    _fields_ = [("base", C.c_void_p), ("batch_stride", C.c_long), ("row_stride", C.c_long),
                ("batch", C.c_int), ("rows_per_batch", C.c_int)]


class ReduceJob(C.Structure):
    """lidbox_reduce_job_t: a wgrad's pending fixed-order slice sum (lidbox_gemm_tn_partial -> lidbox_gemm_nt_carry)"""
    _fields_ = [("partials", C.c_void_p), ("bias_partials", C.c_void_p), ("C", C.c_void_p), ("bias_grad", C.c_void_p),
                ("n", C.c_long), ("ldc", C.c_long), ("splits", C.c_int), ("N", C.c_int), ("accumulate", C.c_int),
                ("nblocks", C.c_uint)]


class WeightShadow(C.Structure):
    """lidbox_weight_shadow_t"""
    _fields_ = [("offset", C.c_long), ("rows", C.c_int), ("cols", C.c_int), ("dst", C.c_void_p), ("ld_dst", C.c_long),
                ("transpose", C.c_int)]


def _load():
    if not os.path.exists(LIB_PATH):
        raise LidboxHipError(
            "liblidbox_hip.so not found at %s -- build it with `python -m lidbox_amd.build` "
            "(lidbox_amd has no CPU fallback)" % LIB_PATH)
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # missing libamdhip64 etc.
        raise LidboxHipError("cannot load %s: %s" % (LIB_PATH, e)) from e
    return lib


lib = _load()

_vp, _i, _l, _f, _sz = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_size_t

_SIGS = {
    "lidbox_hip_abi_version": (_i, []),
    "lidbox_hip_last_error": (C.c_char_p, []),
    "lidbox_ms_to_frames": (_i, [_i, _i]),
    "lidbox_num_frames": (_i, [_i, _i, _i]),
    "lidbox_mel_weight_matrix": (_i, [_i, _i, _i, _f, _f, _vp]),
    "lidbox_hann_window": (_i, [_i, _vp]),
    "lidbox_feat_plan_create": (_i, [_i, _i, _i, _i, _f, _i, _f, _f, _i, _i, C.POINTER(_vp)]),
    "lidbox_feat_plan_destroy": (None, [_vp]),
    "lidbox_feat_plan_channels": (_i, [_vp, _i]),
    "lidbox_feat_plan_is_fused": (_i, [_vp, _i, _vp, _l]),
    "lidbox_extract_features_workspace": (_sz, [_vp, _i, _i, _i, _vp, _l]),
    "lidbox_extract_features_fwd": (_i, [_vp, _i, _vp, _i, _i, _l, _vp, _l, _vp, _sz, _vp]),
    "lidbox_extract_features_fwd_shadow": (_i, [_vp, _i, _vp, _i, _i, _l, _vp, _l, _vp, _vp, _sz, _vp]),
    "lidbox_extract_features_fwd_ex": (_i, [_vp, _i, _vp, _i, _i, _i, _l, _vp, _l, _vp, _vp, _vp, _sz, _vp]),
    "lidbox_extract_features_fwd_pcm16": (_i, [_vp, _i, _vp, _i, _i, _l, _vp, _l, _vp, _vp]),
    "lidbox_cmvn_fwd": (_i, [_vp, _l, _l, _l, _i, _vp, _vp]),
    "lidbox_cmvn_strided_fwd": (_i, [_vp, _l, _l, _l, _l, _i, _vp, _l, _vp]),
    "lidbox_window_norm_fwd": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "lidbox_minmax": (_i, [_vp, _l, _vp, _vp, _vp]),
    "lidbox_feature_scaling_fwd": (_i, [_vp, _l, _vp, _f, _f, _vp, _vp]),
    "lidbox_feature_scaling_axis_fwd": (_i, [_vp, _l, _l, _l, _f, _f, _vp, _vp]),
    "lidbox_log10_fwd": (_i, [_vp, _l, _vp, _vp]),
    "lidbox_power_to_db_fwd": (_i, [_vp, _l, _vp, _f, _f, _vp, _vp]),
    "lidbox_gemm_plan_query": (_i, [_i, _l, _i, _i, _sz, _vp]),
    "lidbox_gemm_last_launches": (_i, [_vp]),
    "lidbox_gemm_last_family": (_i, []),
    "lidbox_gemm_bf16s_last_variant": (_i, [_vp]),
    "lidbox_gemm_plan_stream_tail": (_i, [_l, _i, _i, _i, _sz]),
    "lidbox_gemm_plan_waves": (_i, [_i, _l, _i, _i, _sz]),
    "lidbox_gemm_plan_is_stream_k": (_i, [_i, _l, _i, _i, _sz]),
    "lidbox_gemm_rows_workspace": (_sz, [_l, _i, _i]),
    "lidbox_gemm_nn": (_i, [Rows, _vp, _l, Rows, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "lidbox_gemm_nt": (_i, [Rows, _vp, _l, Rows, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "lidbox_gemm_tn_workspace": (_sz, [_i, _i, _i]),
    "lidbox_gemm_tn": (_i, [Rows, Rows, _vp, _l, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "lidbox_gemm_plan_is_pair": (_i, [_l, _i, _i, _i, _sz, _sz]),
    "lidbox_gemm_nt_tn": (_i, [Rows, _vp, _l, Rows, _i, _i, _i, _vp, _vp, _sz, Rows, _vp, _l, _i, _i, _vp, _vp, _sz, _vp]),
    "lidbox_gemm_tn_partial": (_i, [Rows, Rows, _vp, _l, _i, _i, _i, _vp, _vp, _sz, C.POINTER(ReduceJob), _vp]),
    "lidbox_gemm_nt_carry": (_i, [Rows, _vp, _l, Rows, _i, _i, _i, _vp, _vp, _sz, C.POINTER(ReduceJob), _i, _vp]),
    "lidbox_gemm_nt_tn_carry": (_i, [Rows, _vp, _l, Rows, _i, _i, _i, _vp, _vp, _sz, Rows, _vp, _l, _i, _i, _vp, _vp, _sz,
                                     C.POINTER(ReduceJob), _i, C.POINTER(ReduceJob), _vp]),
    "lidbox_reduce_jobs_run": (_i, [C.POINTER(ReduceJob), _i, _vp]),
    "lidbox_gemm_last_carried": (_i, []),
    "lidbox_zero_job": (_i, [_vp, _l, _l, _i, C.POINTER(ReduceJob)]),
    "lidbox_gemm_bf16_rows_workspace": (_sz, [_l, _i, _i]),
    "lidbox_gemm_bf16_nn": (_i, [Rows, _vp, _l, Rows, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "lidbox_gemm_bf16_nt": (_i, [Rows, _vp, _l, Rows, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "lidbox_gemm_bf16_tn_workspace": (_sz, [_i, _i, _i]),
    "lidbox_gemm_bf16_tn": (_i, [Rows, Rows, _vp, _l, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "lidbox_colsum_workspace": (_sz, [_l, _i]),
    "lidbox_colsum": (_i, [Rows, _i, _vp, _i, _vp, _sz, _vp]),
    "lidbox_stats_pool_fwd": (_i, [_vp, _i, _i, _i, _l, _l, _vp, _vp]),
    "lidbox_stats_pool_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _l, _l, _i, _vp, _vp]),
    "lidbox_softmax_head_workspace": (_sz, [_i, _i, _i]),
    "lidbox_softmax_head_supported": (_i, [_i, _i]),
    "lidbox_softmax_head_fwd_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _f, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "lidbox_stats_pool_bwd_shadow": (_i, [_vp, _vp, _vp, _i, _i, _i, _l, _l, _i, _vp, _vp, _l, _l, _vp]),
    "lidbox_stats_pool_fwd_bf16": (_i, [_vp, _i, _i, _i, _l, _l, _vp, _vp]),
    "lidbox_stats_pool_bwd_bf16": (_i, [_vp, _vp, _vp, _i, _i, _i, _l, _l, _i, _vp, _l, _l, _vp]),
    "lidbox_avg_pool_fwd": (_i, [_vp, _i, _i, _i, _l, _l, _vp, _vp]),
    "lidbox_avg_pool_bwd": (_i, [_vp, _vp, _i, _i, _i, _l, _l, _i, _vp, _vp]),
    "lidbox_signal_chunk_plan": (_i, [_l, _i, _i, _i, _i, _vp]),
    "lidbox_frame_rms": (_i, [_vp, _vp, _vp, _i, _l, _i, _vp, _vp]),
    "lidbox_vad_decisions": (_i, [_vp, _vp, _i, _l, _f, _f, _i, _vp, _vp, _vp, _vp, _vp]),
    "lidbox_vad_scan": (_i, [_vp, _vp, _i, _vp, _vp, _vp]),
    "lidbox_segment_mean": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "lidbox_pcm16_to_f32": (_i, [_vp, _l, _i, _vp, _vp]),
    "lidbox_apply_vad": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _l, _i, _vp, _vp]),
    "lidbox_signal_chunks": (_i, [_vp, _vp, _vp, _vp, _i, _l, _i, _i, _vp, _vp]),
    "lidbox_peak_normalize": (_i, [_vp, _vp, _vp, _i, _f, _vp, _vp]),
    "lidbox_peak_normalize_max": (_i, [_vp, _vp, _vp, _i, _f, _l, _i, _vp, _vp]),
    "lidbox_signal_rms": (_i, [_vp, _vp, _vp, _i, _vp, _vp]),
    "lidbox_snr_mixer": (_i, [_vp, _vp, _vp, _i, _l, _vp, _vp, _vp, _vp]),
    "lidbox_freq_attention_fwd": (_i, [_vp, _vp, _l, _i, _i, _vp, _vp, _vp]),
    "lidbox_freq_attention_bwd": (_i, [_vp, _vp, _vp, _l, _i, _i, _i, _vp, _vp, _vp]),
    "lidbox_log_softmax_fwd": (_i, [_vp, _i, _i, _vp, _vp]),
    "lidbox_nll_fwd_bwd": (_i, [_vp, _vp, _i, _i, _f, _vp, _vp, _vp]),
    "lidbox_l2_normalize_fwd": (_i, [_vp, _i, _i, _vp, _vp]),
    "lidbox_l2_normalize_bwd": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "lidbox_ap_loss_fwd_bwd": (_i, [_vp, _vp, _i, _i, _i, _f, _f, _vp, _vp, _vp]),
    "lidbox_ap_head_fwd_bwd": (_i, [_vp, _vp, _i, _i, _i, _f, _f, _vp, _vp, _vp, _vp, _vp]),
    "lidbox_cavg_update": (_i, [_vp, _vp, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "lidbox_cavg_result": (_i, [_vp, _vp, _vp, _vp, _i, _i, _f, _f, _f, _vp, _vp, _vp]),
    "lidbox_softmax_fwd": (_i, [_vp, _i, _i, _vp, _vp]),
    "lidbox_softmax_nll_fwd_bwd": (_i, [_vp, _vp, _i, _i, _f, _vp, _vp, _vp, _vp]),
    "lidbox_sgd_step": (_i, [_vp, _vp, _vp, _l, _f, _f, _i, _f, _vp, _vp]),
    "lidbox_rmsprop_step": (_i, [_vp, _vp, _vp, _vp, _vp, _l, _f, _f, _f, _f, _i, _f, _vp, _vp]),
    "lidbox_adam_step": (_i, [_vp, _vp, _vp, _vp, _l, _f, _f, _f, _f, _f, _vp, _vp]),
    "lidbox_adam_prepare_job": (_i, [_vp, _f, _f, _f, C.POINTER(ReduceJob)]),
    "lidbox_adam_apply": (_i, [_vp, _vp, _vp, _vp, _l, _f, _f, _f, _f, _vp, _vp]),
    "lidbox_fill": (_i, [_vp, _l, _f, _vp]),
    "lidbox_dropout_rows": (_i, [Rows, _i, _f, C.c_ulonglong, _vp, _vp]),
    "lidbox_mean": (_i, [_vp, _l, _vp, _vp]),
    "lidbox_gemm_bf16s_nt": (_i, [Rows, _vp, _l, Rows, _vp, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "lidbox_gemm_bf16s_tn_workspace": (_sz, [_i, _i, _i]),
    "lidbox_refresh_bf16_weights": (_i, [_vp, _vp, _l, C.POINTER(WeightShadow), _i, _vp]),
    "lidbox_gemm_bf16s_tn": (_i, [Rows, Rows, _vp, _l, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "lidbox_gemm_bf16s_tn_partial": (_i, [Rows, Rows, _vp, _l, _i, _i, _i, _vp, _vp, _sz, C.POINTER(ReduceJob), _vp]),
    "lidbox_gemm_bf16s_nt_carry": (_i, [Rows, _vp, _l, Rows, _vp, _i, _i, _i, _vp, _vp, _sz, C.POINTER(ReduceJob), _i, _vp]),
    "lidbox_gemm_bf16s_last_carried": (_i, []),
    "lidbox_gemm_bf16s_nt_pair_carry": (_i, [Rows, _vp, _l, Rows, _vp, _i, _i, _i, _vp, Rows, _vp, _l, Rows, _vp, _i, _i, _i, _vp,
                                             _vp, _sz, C.POINTER(ReduceJob), _i, _vp]),
    "lidbox_gemm_bf16s_last_pair": (_i, []),
    "lidbox_gemm_bf16s_tn_last_pp": (_i, []),
    "lidbox_gemm_bf16s_tn_last_kres": (_i, []),
    "lidbox_f32_to_bf16": (_i, [_vp, _vp, _l, _vp]),
    "lidbox_bf16_to_f32": (_i, [_vp, _vp, _l, _vp]),
    "lidbox_calibration_spin": (_i, [C.c_double, _vp]),
    "lidbox_transpose_f32_to_bf16": (_i, [_vp, _i, _i, _l, _vp, _l, _vp]),
    "lidbox_scale": (_i, [_vp, _l, _f, _vp]),
    "lidbox_bn_workspace": (_sz, [_l, _i]),
    "lidbox_bn_train_stats": (_i, [_vp, _l, _i, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "lidbox_bn_infer_consts": (_i, [_vp, _vp, _vp, _vp, _f, _i, _vp, _vp, _vp]),
    "lidbox_bn_apply": (_i, [_vp, _l, _i, _vp, _vp, Rows, _vp]),
    "lidbox_bn_bwd": (_i, [_vp, Rows, _l, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "lidbox_neg_acos": (_i, [_vp, _l, _i, _i, _vp, _vp]),
    "lidbox_spatial_dropout": (_i, [_vp, _i, _i, _i, _l, _f, C.c_ulonglong, _vp, _vp, _vp]),
    "lidbox_copy_2d": (_i, [_vp, _sz, _vp, _sz, _sz, _sz, _vp]),
    "lidbox_zero_2d": (_i, [_vp, _sz, _sz, _sz, _vp]),
}

for _name, (_res, _args) in _SIGS.items():
    _fn = getattr(lib, _name)      # AttributeError here = the .so is stale: rebuild
    _fn.restype = _res
    _fn.argtypes = _args

if lib.lidbox_hip_abi_version() != ABI_VERSION:
    raise LidboxHipError("liblidbox_hip.so ABI %d != expected %d: rebuild with `python -m lidbox_amd.build --force`"
                         % (lib.lidbox_hip_abi_version(), ABI_VERSION))


def last_error():
    return lib.lidbox_hip_last_error().decode("utf-8", "replace")


def check(status):
    """Raise on a negative status, mirroring the reference's exception-on-error behaviour
    (tf.debugging.assert_* -> InvalidArgumentError, lidbox/data/tf_utils.py:168-194)."""
    if status != 0:
        msg = last_error()
        if status == -1:
            raise ValueError(msg)
        raise LidboxHipError("status %d: %s" % (status, msg))


def ptr(t):
    """raw device (or host) pointer of a torch tensor / None"""
    return None if t is None else C.c_void_p(t.data_ptr())


def current_stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_gpu_tensor(t, name, dtype=None):
    import torch
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise LidboxHipError("%s is on %s: lidbox_amd runs on the HIP device only (no CPU fallback)"
                             % (name, t.device))
    if dtype is not None and t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    return t
