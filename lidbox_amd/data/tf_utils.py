"""
Counterpart of lidbox/data/tf_utils.py:166-195 `extract_features` (module name kept so that a
lidbox user finds it where they expect it).  For melspectrogram / logmelspectrogram / mfcc the
whole chain waveform -> |STFT|^p -> mel -> ln(.+1e-6) -> DCT runs as ONE fused HIP kernel; the
reference materialises five intermediates for it.
"""
import torch

from .. import _native as nv
from .. import features
from ..features import audio as audio_features

_KIND = {"spectrogram": nv.FEAT_SPECTROGRAM, "db_spectrogram": nv.FEAT_SPECTROGRAM,
         "melspectrogram": nv.FEAT_MEL, "logmelspectrogram": nv.FEAT_LOGMEL, "mfcc": nv.FEAT_MFCC}


def _assert_all_finite(X, message):
    # tf.debugging.assert_all_finite (reference tf_utils.py:173-194)
    if not bool(torch.isfinite(X).all()):
        raise FloatingPointError(message)


def extract_features(signals, sample_rates, feattype, spec_kwargs=None, melspec_kwargs=None, mfcc_kwargs=None,
                     db_spec_kwargs=None, feat_scale_kwargs=None, window_norm_kwargs=None, check_finite=True):
    """reference lidbox/data/tf_utils.py:166-195.  signals [B,N] float32 (or int16: 16-bit PCM read in place) on the HIP
    device, sample_rates [B] (tensor, list or int).  check_finite=False skips the device->host sync the
    reference's asserts imply."""
    if not isinstance(signals, torch.Tensor) or signals.dim() != 2:
        raise ValueError("Input signals for feature extraction must be batches of mono signals without "
                         "channels, i.e. of shape [B, N] where B is batch size and N number of samples.")
    if isinstance(sample_rates, int):
        rates = [sample_rates]
    elif isinstance(sample_rates, (list, tuple)) and sample_rates and all(type(r) is int for r in sample_rates[:1]):
        rates = sample_rates
    else:
        rates = torch.as_tensor(sample_rates).reshape(-1).tolist()
    if rates.count(rates[0]) != len(rates):
        raise ValueError("Different sample rates in a single batch not supported, all signals in the same "
                         "batch should have the same sample rate.")
    if feattype not in _KIND:
        raise ValueError("unknown feature type '%s'" % feattype)
    sample_rate = int(rates[0])
    spec_kwargs = dict(spec_kwargs or {})
    melspec_kwargs = dict(melspec_kwargs or {})
    mfcc_kwargs = dict(mfcc_kwargs or {})
    frame_length = audio_features.ms_to_frames(sample_rate, spec_kwargs.get("frame_length_ms", 25))
    frame_step = audio_features.ms_to_frames(sample_rate, spec_kwargs.get("frame_step_ms", 10))
    plan = audio_features.get_plan(
        sample_rate, frame_length, frame_step, spec_kwargs.get("fft_length", 512), spec_kwargs.get("power", 2.0),
        melspec_kwargs.get("num_mel_bins", 40), melspec_kwargs.get("fmin", 0.0), melspec_kwargs.get("fmax", 8000.0),
        mfcc_kwargs.get("coef_begin", 1), mfcc_kwargs.get("coef_end", 13), device=signals.device)
    # the kernel's store stage folds "a value is not finite" into a 4-byte flag: tf.debugging.assert_all_finite (reference
    # tf_utils.py:173-194) costs one scalar read instead of a pass over the features
    flag = audio_features.FiniteFlag.get(signals.device) if check_finite else None
    X = plan.run(_KIND[feattype], signals, nonfinite=flag)
    if check_finite and not flag.ok():
        raise FloatingPointError(feattype + " failed")
    if feattype == "db_spectrogram":
        X = audio_features.power_to_db(X, **(db_spec_kwargs or {}))
        if check_finite:
            _assert_all_finite(X, "db_spectrogram failed")
    if feat_scale_kwargs:
        X = features.feature_scaling(X, **feat_scale_kwargs)
        if check_finite:
            _assert_all_finite(X, "feature scaling failed")
    if window_norm_kwargs:
        X = features.window_normalization(X, **window_norm_kwargs)
        if check_finite:
            _assert_all_finite(X, "window normalization failed")
    return X
