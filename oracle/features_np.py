"""
oracle.features_np -- numpy restatement of the lidbox feature path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED for values:
TensorFlow is not importable here, so each function restates the reference
line it follows plus the documented semantics of the TF op behind it.

Every function takes `dtype`: np.float64 is the "truth" used for tolerances,
np.float32 follows the reference's float32 op order as closely as numpy allows.

All file:line citations are relative to /root/reference/.
"""
import numpy as np

LOG_EPS = 1e-6  # lidbox/data/tf_utils.py:178


# --------------------------------------------------------------------------- a1
def ms_to_frames(sample_rate, ms):
    """lidbox/features/audio.py:185-189.
    tf.cast(tf.cast(sr, f32) * 1e-3 * tf.cast(ms, f32), i32): float32 arithmetic,
    left to right, then truncation toward zero."""
    x = np.float32(sample_rate) * np.float32(1e-3)
    x = np.float32(x) * np.float32(ms)
    return int(np.int32(np.float32(x)))


# --------------------------------------------------------------------------- a2
def hann_window(window_length, periodic=True, dtype=np.float64):
    """tf.signal.hann_window (window_ops._raised_cosine_window, a=b=0.5), as used
    by tf.signal.stft's default window_fn (periodic=True).  Same denominator rule
    the reference spells out for its own Blackman window at
    lidbox/features/audio.py:206-212: n = L + periodic*even - 1."""
    L = int(window_length)
    if L == 1:
        return np.ones(1, dtype)
    even = 1 - (L % 2)
    n = dtype(L + int(bool(periodic)) * even - 1)
    count = np.arange(L, dtype=dtype)
    cos_arg = dtype(2.0 * np.pi) * count / n
    return (dtype(0.5) - dtype(0.5) * np.cos(cos_arg)).astype(dtype)


def num_frames(num_samples, frame_length, frame_step):
    """tf.signal.frame(pad_end=False): max(0, 1 + (N - L) // S)."""
    if num_samples < frame_length:
        return 0
    return 1 + (num_samples - frame_length) // frame_step


def frame(signals, frame_length, frame_step):
    """tf.signal.frame(signals, L, S, pad_end=False) over the last axis."""
    signals = np.asarray(signals)
    N = signals.shape[-1]
    T = num_frames(N, frame_length, frame_step)
    idx = (np.arange(T)[:, None] * frame_step + np.arange(frame_length)[None, :])
    if T == 0:
        return np.zeros(signals.shape[:-1] + (0, frame_length), signals.dtype)
    return signals[..., idx]


def stft(signals, frame_length, frame_step, fft_length, dtype=np.float64):
    """tf.signal.stft(signals, frame_length, frame_step, fft_length) with the
    default periodic Hann window, pad_end=False: frame -> window -> rfft where
    rfft right-zero-pads (or crops) each windowed frame to fft_length.
    Call site: lidbox/features/audio.py:229."""
    x = np.asarray(signals, dtype=dtype)
    fr = frame(x, frame_length, frame_step) * hann_window(frame_length, True, dtype)
    # numpy's rfft(n=) crops or zero-pads on the right, exactly like tf.signal.rfft
    S = np.fft.rfft(fr.astype(np.float64), n=fft_length, axis=-1)
    return S if dtype == np.float64 else S.astype(np.complex64)


def spectrograms(signals, sample_rate, frame_length_ms=25, frame_step_ms=10,
                 power=2.0, fft_length=512, dtype=np.float64):
    """lidbox/features/audio.py:219-230: |STFT|^power, [B,N] -> [B,T,fft_length//2+1]."""
    frame_length = ms_to_frames(sample_rate, frame_length_ms)
    frame_step = ms_to_frames(sample_rate, frame_step_ms)
    S = stft(signals, frame_length, frame_step, fft_length, dtype)
    return (np.abs(S) ** dtype(power)).astype(dtype)


# --------------------------------------------------------------------------- a3
_MEL_BREAK_FREQUENCY_HERTZ = 700.0   # lidbox/features/mel_ops.py:19
_MEL_HIGH_FREQUENCY_Q = 1127.0       # lidbox/features/mel_ops.py:20


def _linspace(start, stop, num, dtype):
    """lidbox/features/mel_ops.py:11-16 -- NON-endpoint linspace:
    start + (stop - start) * range / num."""
    rng = np.arange(num, dtype=dtype)
    start, stop, num = dtype(start), dtype(stop), dtype(num)
    return (start + (stop - start) * rng / num).astype(dtype)


def _hertz_to_mel(f, dtype):
    """lidbox/features/mel_ops.py:23-25."""
    f = np.asarray(f, dtype=dtype)
    return (dtype(_MEL_HIGH_FREQUENCY_Q) *
            np.log(dtype(1.0) + f / dtype(_MEL_BREAK_FREQUENCY_HERTZ))).astype(dtype)


def linear_to_mel_weight_matrix(num_mel_bins=20, num_spectrogram_bins=129, sample_rate=8000,
                                lower_edge_hertz=125.0, upper_edge_hertz=3800.0,
                                dtype=np.float64, stock_linspace=False):
    """lidbox/features/mel_ops.py:28-75 -> [num_spectrogram_bins, num_mel_bins].
    stock_linspace=True swaps the vendored non-endpoint `_linspace` (:11-16) for the endpoint-inclusive
    tf.linspace of the TensorFlow file the reference says it was copied from (:1-6); everything else is
    the same code.  Only tests use it, to pin the body of this function against independent
    implementations of the stock HTK filterbank (tests/test_oracle.py)."""
    M, F = int(num_mel_bins), int(num_spectrogram_bins)
    _linspace = (lambda a, b, n, dt: np.linspace(dt(a), dt(b), int(n), dtype=dt)) if stock_linspace \
        else globals()["_linspace"]
    nyquist = dtype(sample_rate) / dtype(2.0)                                   # :39
    linear_frequencies = _linspace(0.0, nyquist, F, dtype)[1:]                  # :40-41
    spec_mel = _hertz_to_mel(linear_frequencies, dtype)[:, None]                # :42-43
    edges = _linspace(_hertz_to_mel(lower_edge_hertz, dtype),
                      _hertz_to_mel(upper_edge_hertz, dtype), M + 2, dtype)    # :49-55
    # tf.signal.frame(edges, 3, 1): M triples (lower, center, upper)
    lower = edges[0:M][None, :]
    center = edges[1:M + 1][None, :]
    upper = edges[2:M + 2][None, :]
    with np.errstate(divide="ignore", invalid="ignore"):
        lower_slopes = (spec_mel - lower) / (center - lower)                    # :64-65
        upper_slopes = (upper - spec_mel) / (upper - center)                    # :66-67
        W = np.maximum(dtype(0.0), np.minimum(lower_slopes, upper_slopes))      # :70-71
    return np.pad(W, [[1, 0], [0, 0]]).astype(dtype)                            # :74-75


# --------------------------------------------------------------------------- a4
def linear_to_mel(spectrograms_, sample_rate, num_mel_bins=40, fmin=0.0, fmax=8000.0,
                  dtype=np.float64):
    """lidbox/features/audio.py:247-261: tensordot(S, W, 1)."""
    S = np.asarray(spectrograms_, dtype=dtype)
    W = linear_to_mel_weight_matrix(num_mel_bins, S.shape[2], sample_rate, fmin, fmax, dtype)
    return (S @ W).astype(dtype)


# --------------------------------------------------------------------------- a5/a6
def log_mel(mel, dtype=np.float64):
    """lidbox/data/tf_utils.py:177-179: ln(X + 1e-6)."""
    return np.log(np.asarray(mel, dtype=dtype) + dtype(LOG_EPS)).astype(dtype)


def dct_matrix(num_mel_bins, dtype=np.float64):
    """tf.signal.mfccs_from_log_mel_spectrograms = dct(type=2, norm=None) * rsqrt(2M):
    c_k = sqrt(2/M) * sum_n x_n cos(pi k (2n+1) / (2M)).  Returns D[M(n), M(k)]."""
    M = int(num_mel_bins)
    n = np.arange(M, dtype=np.float64)[:, None]
    k = np.arange(M, dtype=np.float64)[None, :]
    D = 2.0 * np.cos(np.pi * k * (2.0 * n + 1.0) / (2.0 * M)) / np.sqrt(2.0 * M)
    return D.astype(dtype)


def mfccs_from_log_mel(logmel, coef_begin=1, coef_end=13, dtype=np.float64):
    """lidbox/data/tf_utils.py:180-185."""
    X = np.asarray(logmel, dtype=dtype)
    D = dct_matrix(X.shape[-1], dtype)
    return (X @ D)[..., coef_begin:coef_end].astype(dtype)


# --------------------------------------------------------------------------- a7
def divide_no_nan(x, y):
    """tf.math.divide_no_nan: 0 where y == 0."""
    x, y = np.broadcast_arrays(x, y)
    out = np.zeros_like(x)
    np.divide(x, y, out=out, where=(y != 0))
    return out


def cmn(X, axis=1, dtype=np.float64):
    """lidbox/features/__init__.py:12-20."""
    X = np.asarray(X, dtype=dtype)
    return X - X.mean(axis=axis, keepdims=True)


def cmvn(X, axis=1, dtype=np.float64):
    """lidbox/features/__init__.py:22-32: reduce_std is the POPULATION std of X."""
    X = np.asarray(X, dtype=dtype)
    return divide_no_nan(cmn(X, axis, dtype), X.std(axis=axis, keepdims=True))


# --------------------------------------------------------------------------- a8
def window_normalization(X, axis=1, window_len=-1, normalize_variance=True, dtype=np.float64):
    """lidbox/features/__init__.py:35-67.  The sliding branch always pads and
    compares on dim 1 (the code, not the comment: mode="REFLECT" does not repeat
    the edge sample)."""
    X = np.asarray(X, dtype=dtype)
    if window_len == -1 or X.shape[1] <= window_len:
        return cmvn(X, axis, dtype) if normalize_variance else cmn(X, axis, dtype)
    w = int(window_len)
    pad = [(0, 0), (w // 2, w // 2 - 1 + (w & 1)), (0, 0)]
    Xp = np.pad(X, pad, mode="reflect")
    # tf.signal.frame(Xp, w, 1, axis=axis) -> windows along a new axis after `axis`
    T = X.shape[axis]
    idx = np.arange(T)[:, None] + np.arange(w)[None, :]
    windows = np.take(Xp, idx, axis=axis)           # [..., T, w, ...]
    assert windows.shape[axis] == X.shape[axis]
    out = X - windows.mean(axis=axis + 1)
    if normalize_variance:
        out = divide_no_nan(out, windows.std(axis=axis + 1))
    return out.astype(dtype)


# --------------------------------------------------------------------------- a9
def feature_scaling(X, min, max, axis=None, dtype=np.float64):
    """lidbox/features/__init__.py:5-9."""
    X = np.asarray(X, dtype=dtype)
    Xmin = X.min(axis=axis, keepdims=True)
    Xmax = X.max(axis=axis, keepdims=True)
    return dtype(min) + dtype(max - min) * divide_no_nan(X - Xmin, Xmax - Xmin)


# --------------------------------------------------------------------------- a10
def log10(x, dtype=np.float64):
    """lidbox/features/audio.py:162-164: ln(x)/ln(10)."""
    x = np.asarray(x, dtype=dtype)
    return (np.log(x) / np.log(dtype(10.0))).astype(dtype)


def power_to_db(S, amin=1e-10, top_db=80.0, dtype=np.float64):
    """lidbox/features/audio.py:167-174: factor 20, max over the WHOLE batch."""
    S = np.asarray(S, dtype=dtype)
    db = dtype(20.0) * (log10(np.maximum(dtype(amin), S), dtype) -
                        log10(np.maximum(dtype(amin), S.max()), dtype))
    return np.maximum(db, db.max() - dtype(top_db)).astype(dtype)


def db_to_power(S, dtype=np.float64):
    """lidbox/features/audio.py:177-181."""
    return np.power(dtype(10.0), np.asarray(S, dtype=dtype) / dtype(20.0))


def fft_frequencies(sample_rate, n_fft, dtype=np.float64):
    """lidbox/features/audio.py:151-159: linspace(0, sr//2, 1 + n_fft//2) (endpoint)."""
    return np.linspace(0.0, float(sample_rate // 2), 1 + n_fft // 2).astype(dtype)


# --------------------------------------------------------------------------- a11
def extract_features(signals, sample_rates, feattype, spec_kwargs=None, melspec_kwargs=None,
                     mfcc_kwargs=None, db_spec_kwargs=None, feat_scale_kwargs=None,
                     window_norm_kwargs=None, dtype=np.float64):
    """lidbox/data/tf_utils.py:166-195.  The call at :175 names a function that
    does not exist (`melspectrograms`); the intent -- linear_to_mel(X,
    sample_rate=..., **melspec_kwargs) -- is what is restated here."""
    signals = np.asarray(signals)
    if signals.ndim != 2:
        raise ValueError("signals must be [B, N]")                              # :168
    sample_rates = np.asarray(sample_rates).reshape(-1)
    if not np.all(sample_rates == sample_rates[0]):
        raise ValueError("different sample rates in one batch")                 # :169
    sr = int(sample_rates[0])
    X = spectrograms(signals, sr, dtype=dtype, **(spec_kwargs or {}))           # :172

    def _finite(X, what):
        if not np.all(np.isfinite(X)):
            raise FloatingPointError(what + " failed")
    _finite(X, "spectrogram")
    if feattype in ("melspectrogram", "logmelspectrogram", "mfcc"):
        X = linear_to_mel(X, sr, dtype=dtype, **(melspec_kwargs or {}))         # :175
        _finite(X, "melspectrogram")
        if feattype in ("logmelspectrogram", "mfcc"):
            X = log_mel(X, dtype)                                               # :178
            _finite(X, "logmelspectrogram")
            if feattype == "mfcc":
                mk = mfcc_kwargs or {}
                X = mfccs_from_log_mel(X, mk.get("coef_begin", 1), mk.get("coef_end", 13), dtype)
                _finite(X, "mfcc")
    elif feattype in ("db_spectrogram",):
        X = power_to_db(X, dtype=dtype, **(db_spec_kwargs or {}))               # :187
        _finite(X, "db_spectrogram")
    if feat_scale_kwargs:
        X = feature_scaling(X, dtype=dtype, **feat_scale_kwargs)                # :190
        _finite(X, "feature scaling")
    if window_norm_kwargs:
        X = window_normalization(X, dtype=dtype, **window_norm_kwargs)          # :193
        _finite(X, "window normalization")
    return X.astype(dtype)


# --------------------------------------------------------------------------- I/O
def read_wav_pcm16(path):
    """tf.audio.decode_wav semantics for 16-bit PCM (lidbox/features/audio.py:17-23):
    int16 / 32768 -> float32 in [-1, 1), channels averaged."""
    import wave
    with wave.open(path, "rb") as f:
        assert f.getsampwidth() == 2
        nch, sr, n = f.getnchannels(), f.getframerate(), f.getnframes()
        raw = np.frombuffer(f.readframes(n), dtype="<i2").astype(np.float32) / np.float32(32768.0)
    return raw.reshape(-1, nch).mean(axis=1).astype(np.float32), sr
