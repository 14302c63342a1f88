"""
Counterpart of lidbox/features/mel_ops.py (reference :28-75).  The matrix is a constant of
(num_mel_bins, num_spectrogram_bins, sample_rate, lower, upper); it is built on the host by
liblidbox_hip.so in float32 op order -- including the reference's NON-endpoint `_linspace`
(:11-16) -- and cached per device.
"""
import threading

import numpy as np
import torch

from .. import _native as nv

_cache = {}
_lock = threading.Lock()


def linear_to_mel_weight_matrix_host(num_mel_bins=20, num_spectrogram_bins=129, sample_rate=8000,
                                     lower_edge_hertz=125.0, upper_edge_hertz=3800.0):
    """numpy float32 [num_spectrogram_bins, num_mel_bins]."""
    W = np.zeros((int(num_spectrogram_bins), int(num_mel_bins)), np.float32)
    nv.check(nv.lib.lidbox_mel_weight_matrix(int(num_mel_bins), int(num_spectrogram_bins), int(sample_rate),
                                             float(lower_edge_hertz), float(upper_edge_hertz), W.ctypes.data))
    return W


def linear_to_mel_weight_matrix(num_mel_bins=20, num_spectrogram_bins=129, sample_rate=8000,
                                lower_edge_hertz=125.0, upper_edge_hertz=3800.0, dtype=torch.float32,
                                device=None, name=None):
    """Same signature as the reference; returns a device tensor [F, M]."""
    if dtype != torch.float32:
        raise TypeError("only float32 is supported")
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    key = (str(device), int(num_mel_bins), int(num_spectrogram_bins), int(sample_rate),
           float(lower_edge_hertz), float(upper_edge_hertz))
    with _lock:
        W = _cache.get(key)
        if W is None:
            W = torch.from_numpy(linear_to_mel_weight_matrix_host(*key[1:])).to(device)
            _cache[key] = W
    return W
