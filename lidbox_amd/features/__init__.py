"""
Counterpart of lidbox/features/__init__.py: feature_scaling, cmn, cmvn, window_normalization
with the reference's signatures (file:line cited per function), over torch tensors on the HIP
device.  Arithmetic runs in liblidbox_hip.so; there is no CPU fallback.
"""
import torch

from .. import _native as nv


def _as_outer_r_inner(X, axis):
    axis = axis % X.dim()
    outer = 1
    for d in X.shape[:axis]:
        outer *= d
    inner = 1
    for d in X.shape[axis + 1:]:
        inner *= d
    return outer, X.shape[axis], inner


def _cmvn(X, axis, normalize_variance):
    X = nv.require_gpu_tensor(X, "X", torch.float32).contiguous()
    out = torch.empty_like(X)
    if X.numel() == 0:
        return out
    outer, R, inner = _as_outer_r_inner(X, axis)
    with torch.cuda.device(X.device):
        nv.check(nv.lib.lidbox_cmvn_fwd(nv.ptr(X), outer, R, inner, int(bool(normalize_variance)), nv.ptr(out),
                                        nv.current_stream()))
    return out


def feature_scaling(X, min, max, axis=None):
    """reference lidbox/features/__init__.py:5-9.  axis=None (the form the pipeline uses, tf_utils.py:189-190): the fused
    min/max + rescale kernels over the whole tensor; an explicit axis: one kernel that takes min / max over that axis
    (lidbox_feature_scaling_axis_fwd); a tuple of axes is reduced as one axis (after a permutation when they are not adjacent)."""
    X = nv.require_gpu_tensor(X, "X", torch.float32)
    if axis is not None:
        X = X.contiguous()
        out = torch.empty_like(X)
        if isinstance(axis, int):
            outer, R, inner = _as_outer_r_inner(X, axis)
        else:
            # a tuple / list of axes (tf.math.reduce_min accepts one): adjacent axes are one reduced axis of the
            # contiguous tensor; anything else is reduced after a permutation that makes them adjacent
            axes = sorted({int(a) % X.dim() for a in axis})
            if not axes:
                raise ValueError("feature_scaling: empty axis tuple")
            if axes != list(range(axes[0], axes[-1] + 1)):
                keep = [d for d in range(X.dim()) if d not in axes]
                perm = keep[:axes[0]] + axes + keep[axes[0]:]
                inv = [perm.index(d) for d in range(X.dim())]
                res = feature_scaling(X.permute(perm).contiguous(), min, max, axis=tuple(range(axes[0], axes[0] + len(axes))))
                return res.permute(inv).contiguous()
            outer, _, _ = _as_outer_r_inner(X, axes[0])
            _, _, inner = _as_outer_r_inner(X, axes[-1])
            R = 1
            for a in axes:
                R *= X.shape[a]
        if X.numel() == 0:
            return out
        with torch.cuda.device(X.device):
            nv.check(nv.lib.lidbox_feature_scaling_axis_fwd(nv.ptr(X), outer, R, inner, float(min), float(max), nv.ptr(out),
                                                            nv.current_stream()))
        return out
    X = X.contiguous()
    out = torch.empty_like(X)
    if X.numel() == 0:
        return out
    from .audio import _minmax
    mm = _minmax(X)
    with torch.cuda.device(X.device):
        nv.check(nv.lib.lidbox_feature_scaling_fwd(nv.ptr(X), X.numel(), nv.ptr(mm), float(min), float(max),
                                                   nv.ptr(out), nv.current_stream()))
    return out


def cmn(X, axis=1):
    """reference lidbox/features/__init__.py:12-20."""
    return _cmvn(X, axis, False)


def cmvn(X, axis=1):
    """reference lidbox/features/__init__.py:22-32 (population std, divide_no_nan)."""
    return _cmvn(X, axis, True)


def window_normalization(X, axis=1, window_len=-1, normalize_variance=True):
    """reference lidbox/features/__init__.py:35-67."""
    X = nv.require_gpu_tensor(X, "X", torch.float32)
    if X.dim() != 3:
        raise ValueError("X must be [B, T, C]")
    if window_len == -1 or X.shape[1] <= window_len:
        return cmvn(X, axis=axis) if normalize_variance else cmn(X, axis=axis)
    if axis != 1:
        raise ValueError("sliding window normalization is defined on axis=1 (the reference pads dim 1)")
    X = X.contiguous()
    out = torch.empty_like(X)
    if X.numel() == 0:
        return out
    B, T, C = X.shape
    with torch.cuda.device(X.device):
        nv.check(nv.lib.lidbox_window_norm_fwd(nv.ptr(X), B, T, C, int(window_len), int(bool(normalize_variance)),
                                               nv.ptr(out), nv.current_stream()))
    return out
