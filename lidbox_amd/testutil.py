"""
Synthetic inputs for tests and bench (counterpart of lidbox/testutil.py:9-26 in the
reference: peak_normalize + noisy_sinewave), restated with a seeded generator so that
every rank / every run sees the same batch.  Recipe = SURVEY.md section 8(d).
"""
import numpy as np


def peak_normalize(signal, dBFS):
    """reference lidbox/testutil.py:9-12, row-wise for a batch."""
    level = 10.0 ** (dBFS / 20.0)
    peak = np.max(np.abs(signal), axis=-1, keepdims=True)
    return level * (signal / peak)


def synthetic_batch(batch_size, num_labels=4, sample_rate=16000, duration_s=2.0, seed=1234):
    """Deterministic batch of noisy tones, one tone frequency per label:
    x_b[n] = sin(2 pi 100 (1 + y_b mod 8) n / sr) + g^2, g ~ N(0, 0.1), peak-normalised
    to -3 dBFS (reference lidbox/testutil.py:23-26).  Returns (signals f32 [B,N], labels i32 [B])."""
    rng = np.random.Generator(np.random.PCG64(seed))
    n = int(round(sample_rate * duration_s))
    labels = rng.integers(0, num_labels, size=batch_size, dtype=np.int32)
    t = np.arange(n, dtype=np.float64) / sample_rate
    freqs = 100.0 * (1 + (labels % 8)).astype(np.float64)
    tone = np.sin(2.0 * np.pi * freqs[:, None] * t[None, :])
    noise = rng.normal(0.0, 0.1, size=(batch_size, n)) ** 2
    return peak_normalize(tone + noise, -3.0).astype(np.float32), labels
