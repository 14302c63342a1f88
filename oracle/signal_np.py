"""
oracle.signal_np -- numpy restatement of the lidbox signal steps that sit immediately before the feature
kernel: RMS, peak normalisation, run-length encoding, energy VAD, silence removal, VAD
application, fixed-length chunking and the SNR mixer.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Pinned by the reference's own tests for these functions
(tests/test_features_audio.py:49-67, 157-191: dBFS_to_linear, peak_normalize bound, root_mean_square 1e-5,
run_length_encoding known answer, VAD all-speech on the WAV fixtures / all-silence on zeros, remove_silence
shape laws) -- restated in tests/test_oracle_signal.py -- and by `numpy_snr_mixer` (audio.py:100-121), the
reference's own numpy twin of snr_mixer.  VALUES of the TF versions: parity unpinned (no TensorFlow here).

All file:line citations are relative to /root/reference/lidbox/.
"""
import numpy as np

from .features_np import ms_to_frames


def dBFS_to_linear(level):
    """features/audio.py:49-51"""
    return np.float32(10.0) ** (np.float32(level) / np.float32(20.0))


def peak_normalize(signal, dBFS=0.0):
    """features/audio.py:57-59"""
    signal = np.asarray(signal)
    return dBFS_to_linear(dBFS) * (signal / np.max(np.abs(signal)))


def root_mean_square(x, axis=-1):
    """features/audio.py:266-270"""
    x = np.asarray(x)
    return np.sqrt(np.mean(np.square(np.abs(x)), axis=axis))


def run_length_encoding(v):
    """features/audio.py:275-283 -> (start positions, lengths) of the runs of equal values"""
    v = np.asarray(v)
    if v.size == 0:
        return np.zeros(0, np.int64), np.zeros(0, np.int64)
    i = np.concatenate(([-1], np.nonzero(v[1:] != v[:-1])[0], [v.size - 1]))
    pos = np.concatenate(([0], np.cumsum(i[1:] - i[:-1])))
    return pos[:-1], pos[1:] - pos[:-1]


def invert_too_short_consecutive_false(mask, min_length):
    """features/audio.py:289-297: runs of False shorter than min_length become True"""
    mask = np.asarray(mask, bool)
    if min_length == 0 or mask.size == 0:
        return mask
    pos, lengths = run_length_encoding(mask.astype(np.int32))
    true_or_too_short = np.logical_or(mask[pos], lengths < min_length)
    return np.repeat(true_or_too_short, lengths)


def frame_nonoverlapping(signal, frame_length):
    """tf.signal.frame(signal, L, L) with pad_end=False: [N // L, L]; the tail is dropped"""
    signal = np.asarray(signal)
    n = signal.shape[0] // frame_length if frame_length > 0 else 0
    return signal[:n * frame_length].reshape(n, frame_length)


def framewise_rms_energy_vad_decisions(signal, sample_rate, frame_step_ms, min_non_speech_ms=0, strength=0.05,
                                       min_rms_threshold=1e-3, dtype=np.float64):
    """features/audio.py:308-329 (time_axis=0).  dtype=np.float32 follows the reference's arithmetic type."""
    signal = np.asarray(signal, dtype)
    frame_step = ms_to_frames(sample_rate, frame_step_ms)
    frames = frame_nonoverlapping(signal, frame_step)                                  # :314
    rms = root_mean_square(frames, axis=1).astype(dtype)                               # :317
    if rms.size == 0:
        return np.zeros(0, bool)
    mean_rms = np.mean(rms, dtype=dtype)                                               # :318
    threshold = dtype(strength) * np.maximum(dtype(min_rms_threshold), mean_rms)       # :321
    vad = rms > threshold                                                              # :322
    min_non_speech_frames = int(ms_to_frames(sample_rate, min_non_speech_ms) / frame_step)   # :325 (int64 cast truncates)
    return invert_too_short_consecutive_false(vad, min_non_speech_frames)              # :326


def remove_silence(signal, rate, window_ms=10, min_non_speech_ms=300, dtype=np.float64):
    """features/audio.py:337-353"""
    window_frames = (window_ms * rate) // 1000
    vad_1 = framewise_rms_energy_vad_decisions(signal, rate, window_ms, min_non_speech_ms, strength=0.1, dtype=dtype)
    windows = frame_nonoverlapping(np.asarray(signal), window_frames)
    return windows[vad_1].reshape(-1)


def apply_vad(signal, sample_rate, vad_frame_length_ms, vad_is_speech):
    """data/steps.py:191-198: keep the VAD frames marked as speech, flattened"""
    sec = np.float32(1e-3) * np.float32(vad_frame_length_ms)
    vad_frame_length = int(np.float32(sample_rate) * sec)
    frames = frame_nonoverlapping(np.asarray(signal), vad_frame_length)
    return frames[np.asarray(vad_is_speech, bool)].reshape(-1)


def signal_chunk_plan(num_samples, sample_rate, length_ms, step_ms, max_pad_ms=0):
    """data/steps.py:586-588, 604-614: (chunk_length, chunk_step, padded signal length, number of chunks)"""
    sr = np.float32(sample_rate)
    chunk_length = int(sr * np.float32(1e-3 * length_ms))
    chunk_step = int(sr * np.float32(1e-3 * step_ms))
    max_pad = int(sr * np.float32(1e-3 * max_pad_ms))
    n = int(num_samples)
    num_full_chunks = max(0, 1 + (n - chunk_length) // chunk_step)                     # :607
    last_chunk_length = n - num_full_chunks * chunk_step                               # :610
    if last_chunk_length < chunk_length and chunk_length <= last_chunk_length + max_pad:   # :611
        n = n + chunk_length - last_chunk_length                                       # :612
    num_chunks = max(0, 1 + (n - chunk_length) // chunk_step) if n >= chunk_length else 0   # tf.signal.frame, :614
    return chunk_length, chunk_step, n, num_chunks


def create_signal_chunks(signal, sample_rate, length_ms, step_ms, max_pad_ms=0):
    """data/steps.py:600-614 -> [num_chunks, chunk_length]"""
    signal = np.asarray(signal)
    L, S, n, nc = signal_chunk_plan(signal.shape[0], sample_rate, length_ms, step_ms, max_pad_ms)
    padded = np.concatenate([signal, np.zeros(n - signal.shape[0], signal.dtype)])
    idx = np.arange(nc)[:, None] * S + np.arange(L)[None, :]
    return padded[idx] if nc > 0 else np.zeros((0, L), signal.dtype)


def snr_mixer(clean, noise, snr, dtype=np.float64):
    """features/audio.py:128-148 (TF version; numpy twin at :100-121)"""
    clean, noise = np.asarray(clean, dtype), np.asarray(noise, dtype)
    assert clean.size == noise.size                                                    # :132
    lvl25 = dtype(10.0) ** dtype(-25.0 / 20.0)
    clean_norm = (lvl25 / root_mean_square(clean)) * clean                             # :134-135
    rmsclean = root_mean_square(clean_norm)
    noise_norm = (lvl25 / root_mean_square(noise)) * noise                             # :138-139
    rmsnoise = root_mean_square(noise_norm)
    level = dtype(10.0) ** (dtype(snr) / dtype(20.0))                                  # :143
    noisescalar = np.sqrt(rmsclean / level / rmsnoise)                                 # :144
    noisenewlevel = noisescalar * noise_norm
    return clean_norm, noisenewlevel, clean_norm + noisenewlevel


def numpy_snr_mixer_reference_twin(clean, noise, snr):
    """features/audio.py:100-121 restated verbatim in meaning (the reference's own numpy version)"""
    rmsclean = (clean ** 2).mean() ** 0.5
    clean = clean * (10 ** (-25 / 20) / rmsclean)
    rmsclean = (clean ** 2).mean() ** 0.5
    rmsnoise = (noise ** 2).mean() ** 0.5
    noise = noise * (10 ** (-25 / 20) / rmsnoise)
    rmsnoise = (noise ** 2).mean() ** 0.5
    noisescalar = np.sqrt(rmsclean / (10 ** (snr / 20)) / rmsnoise)
    noisenewlevel = noise * noisescalar
    return clean, noisenewlevel, clean + noisenewlevel
