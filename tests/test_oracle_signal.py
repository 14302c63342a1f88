"""
Pins oracle/signal_np.py (the CPU restatement of the lidbox signal steps) to every property the
reference's own tests hold for these functions (reference tests/test_features_audio.py) and to the reference's
numpy twin of the SNR mixer.  CPU only.
"""
import glob
import os

import numpy as np
import pytest

from oracle import features_np as fo
from oracle import signal_np as so

AUDIO = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "audio", "*.wav")))


def test_dbfs_to_linear():
    """reference tests/test_features_audio.py:49-57"""
    for i, level in enumerate(range(0, 200, 20)):
        assert abs(so.dBFS_to_linear(level) - 10.0 ** i) < 1e-6 * 10.0 ** i or abs(so.dBFS_to_linear(level) - 10.0 ** i) < 1e-6


def test_peak_normalize_bound():
    """reference tests/test_features_audio.py:59-66"""
    rng = np.random.default_rng(0)
    for path in AUDIO:
        s, r = fo.read_wav_pcm16(path)
        s1 = s + rng.normal(0, 10, s.shape)
        for level in range(0, -10, -1):
            s2 = so.peak_normalize(s1, dBFS=level)
            assert not np.isnan(s2).any()
            assert np.max(np.abs(s2)) <= so.dBFS_to_linear(level) * (1 + 1e-6)


def test_root_mean_square():
    """reference tests/test_features_audio.py:157-163"""
    rng = np.random.default_rng(1)
    for _ in range(100):
        x = rng.normal(0, 5, size=rng.integers(1, 10, size=2))
        rms1 = np.sqrt(np.mean(np.square(np.abs(x)), axis=-1))
        assert np.abs(rms1 - so.root_mean_square(x.astype(np.float32), axis=-1)).max() < 1e-5


def test_run_length_encoding_known_answer():
    """reference tests/test_features_audio.py:166-169"""
    pos, length = so.run_length_encoding(np.array([1, 1, 1, 2, 2, 2, 3, 4, 5, 6, 6, 7]))
    assert (pos == np.array([0, 3, 6, 7, 8, 9, 11])).all()
    assert (length == np.array([3, 3, 1, 1, 1, 2, 1])).all()


def test_invert_too_short_consecutive_false():
    m = np.array([0, 1, 0, 0, 1, 0, 0, 0, 1, 1, 0], bool)
    assert (so.invert_too_short_consecutive_false(m, 0) == m).all()
    assert (so.invert_too_short_consecutive_false(m, 3).astype(int) == [1, 1, 1, 1, 1, 0, 0, 0, 1, 1, 1]).all()
    assert so.invert_too_short_consecutive_false(m, 100).all()


def test_vad_decisions_on_fixtures():
    """reference tests/test_features_audio.py:175-181"""
    assert AUDIO
    for path in AUDIO:
        s, r = fo.read_wav_pcm16(path)
        for dt in (np.float64, np.float32):
            vad = so.framewise_rms_energy_vad_decisions(s, r, 25, dtype=dt)
            assert vad.shape == (len(s) // 400,) and vad.all()
    assert not so.framewise_rms_energy_vad_decisions(np.zeros(3 * 16000), 16000, 25).any()


def test_remove_silence_on_fixtures():
    """reference tests/test_features_audio.py:183-191"""
    for path in AUDIO:
        s, r = fo.read_wav_pcm16(path)
        s1 = so.remove_silence(s, r)
        assert not np.isnan(s1).any() and s1.shape == s.shape
    assert so.remove_silence(np.zeros(3 * 16000), 16000).size == 0


def test_apply_vad_and_chunks():
    rng = np.random.default_rng(2)
    s = rng.standard_normal(1000).astype(np.float32)
    dec = rng.random(1000 // 160) > 0.4
    out = so.apply_vad(s, 16000, 10, dec)
    assert out.size == dec.sum() * 160
    assert (out[:160] == s[np.argmax(dec) * 160:][:160]).all()
    # chunk plan (data/steps.py:604-614): 1 s chunks, 0.5 s step, with / without padding allowance
    for n, pad_ms, want in [(16000, 0, 1), (23999, 0, 1), (24000, 0, 2), (23000, 500, 2), (23000, 50, 1),
                            (8000, 0, 0), (8000, 500, 1), (7999, 500, 0), (0, 0, 0)]:
        L, S, _, nc = so.signal_chunk_plan(n, 16000, 1000, 500, pad_ms)
        assert (L, S, nc) == (16000, 8000, want), (n, pad_ms, nc)
        ch = so.create_signal_chunks(np.arange(n, dtype=np.float32) + 1, 16000, 1000, 500, pad_ms)
        assert ch.shape == (want, 16000)
        for c in range(want):
            ref = np.arange(c * 8000, c * 8000 + 16000) + 1.0
            ref[ref > n] = 0
            assert (ch[c] == ref).all()


@pytest.mark.parametrize("snr", [-5.0, 0.0, 12.5])
def test_snr_mixer_matches_reference_numpy_twin(snr):
    """features/audio.py:100-121 (numpy) and :128-148 (TF) are meant to be the same computation"""
    rng = np.random.default_rng(3)
    clean, noise = rng.standard_normal(4000) * 0.3, rng.standard_normal(4000) * 0.02
    a = so.snr_mixer(clean, noise, snr)
    b = so.numpy_snr_mixer_reference_twin(clean, noise, snr)
    for x, y in zip(a, b):
        assert np.abs(x - y).max() < 1e-12
    # the mixture has the requested SNR in the reference's (amplitude-ratio under a square root) sense
    assert abs(so.root_mean_square(a[0]) - 10 ** (-25 / 20)) < 1e-12
