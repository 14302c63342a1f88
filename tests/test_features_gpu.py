"""
GPU parity: HIP feature path (through the C ABI) vs the oracle on identical inputs.

Tolerances (fp32 GPU vs float64 oracle), from SURVEY.md section 8(c) / BASELINE.json north_star:
  log-mel max-abs <= 1e-3 (log domain), MFCC max-abs <= 1e-3, spectrogram/mel relative 2e-5
  of the per-utterance maximum.
"""
import numpy as np
import pytest
import torch

from oracle import features_np as fo

pytestmark = pytest.mark.gpu

LOGMEL_TOL = 1e-3
MFCC_TOL = 1e-3


def _dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


@pytest.fixture(scope="module")
def synth():
    from lidbox_amd.testutil import synthetic_batch
    return synthetic_batch(6, num_labels=4)


def test_ms_to_frames_grid():
    from lidbox_amd.features import audio
    for sr in range(1000, 60000, 1000):
        for ms in range(1, 5000, 100):
            assert audio.ms_to_frames(sr, ms) == (sr // 1000) * ms


def test_fused_logmel_matches_oracle(synth):
    from lidbox_amd.data import tf_utils
    sig, _ = synth
    ref = fo.extract_features(sig, [16000] * len(sig), "logmelspectrogram")
    got = tf_utils.extract_features(_dev(sig), [16000] * len(sig), "logmelspectrogram").cpu().numpy()
    assert got.shape == ref.shape == (6, 198, 40)
    assert np.isfinite(got).all()
    err = np.abs(got - ref).max()
    assert err <= LOGMEL_TOL, err


def test_fused_all_kinds_match_oracle(synth):
    from lidbox_amd.data import tf_utils
    sig, _ = synth
    sr = [16000] * len(sig)
    x = _dev(sig)
    for kind, rel in (("spectrogram", 2e-5), ("melspectrogram", 2e-5)):
        ref = fo.extract_features(sig, sr, kind)
        got = tf_utils.extract_features(x, sr, kind).cpu().numpy()
        assert got.shape == ref.shape
        scale = np.abs(ref).max(axis=(1, 2), keepdims=True)
        assert (np.abs(got - ref) / scale).max() <= rel, kind
    ref = fo.extract_features(sig, sr, "mfcc")
    got = tf_utils.extract_features(x, sr, "mfcc").cpu().numpy()
    assert got.shape == ref.shape == (6, 198, 12)
    assert np.abs(got - ref).max() <= MFCC_TOL
    ref = fo.extract_features(sig, sr, "mfcc", mfcc_kwargs=dict(coef_begin=0, coef_end=20),
                              window_norm_kwargs=dict(window_len=-1, normalize_variance=True))
    got = tf_utils.extract_features(x, sr, "mfcc", mfcc_kwargs=dict(coef_begin=0, coef_end=20),
                                    window_norm_kwargs=dict(window_len=-1, normalize_variance=True)).cpu().numpy()
    assert np.abs(got - ref).max() <= 2e-3


def test_reference_wav_fixtures(wav_paths):
    """the reference's own 16 kHz fixtures (3 s -> T = 298), log-mel and MFCC+CMVN"""
    from lidbox_amd.data import tf_utils
    sigs = np.stack([fo.read_wav_pcm16(p)[0] for p in wav_paths])
    sr = [16000] * len(sigs)
    ref = fo.extract_features(sigs, sr, "logmelspectrogram")
    got = tf_utils.extract_features(_dev(sigs), sr, "logmelspectrogram").cpu().numpy()
    assert got.shape == (5, 298, 40)
    assert np.abs(got - ref).max() <= LOGMEL_TOL
    golden = np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "features_wav.npz"))
    assert np.abs(got - golden["logmel"]).max() <= LOGMEL_TOL


def test_ragged_and_edge_shapes():
    """T not a multiple of 8, a single frame, N < frame_length (T = 0), B = 1, unaligned rows"""
    from lidbox_amd.data import tf_utils
    rng = np.random.default_rng(0)
    for n in (400, 559, 560, 1000, 4001, 16000 + 3):
        sig = rng.standard_normal((3, n)).astype(np.float32) * 0.1
        ref = fo.extract_features(sig, [16000] * 3, "logmelspectrogram")
        got = tf_utils.extract_features(_dev(sig), [16000] * 3, "logmelspectrogram").cpu().numpy()
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() <= LOGMEL_TOL, n
    sig = rng.standard_normal((2, 399)).astype(np.float32)
    assert tf_utils.extract_features(_dev(sig), [16000] * 2, "logmelspectrogram").shape == (2, 0, 40)
    # strided (non-dense) batch rows exercise sig_stride
    big = _dev(rng.standard_normal((4, 5000)).astype(np.float32))
    view = big[:, :4000]
    ref = fo.extract_features(view.cpu().numpy(), [16000] * 4, "logmelspectrogram")
    got = tf_utils.extract_features(view, [16000] * 4, "logmelspectrogram").cpu().numpy()
    assert np.abs(got - ref).max() <= LOGMEL_TOL


def test_generic_path_other_fft_lengths(wav_paths):
    """reference test_spectrograms grid (tests/test_features_audio.py:131-145): shape law + values"""
    from lidbox_amd.features import audio
    s, r = fo.read_wav_pcm16(wav_paths[0])
    s = s[:16000]
    x = _dev(s[None])
    for len_ms in (20, 60, 100):
        for n_fft in (256, 512, 1024, 2048):
            if n_fft < audio.ms_to_frames(r, len_ms):
                continue
            step_ms = len_ms // 2
            got = audio.spectrograms(x, r, frame_length_ms=len_ms, frame_step_ms=step_ms, fft_length=n_fft)[0]
            got = got.cpu().numpy()
            ref = fo.spectrograms(s[None], r, frame_length_ms=len_ms, frame_step_ms=step_ms, fft_length=n_fft)[0]
            assert not np.isnan(got).any()
            assert got.shape[0] == s.shape[0] // audio.ms_to_frames(r, step_ms) - 1
            assert got.shape[1] == n_fft // 2 + 1
            assert np.abs(got - ref).max() <= 2e-5 * np.abs(ref).max(), (len_ms, n_fft)
    # power != 2 and a different sample rate
    got = audio.spectrograms(x, r, power=1.0).cpu().numpy()
    ref = fo.spectrograms(s[None], r, power=1.0)
    assert np.abs(got - ref).max() <= 2e-5 * np.abs(ref).max()
    got = audio.spectrograms(x, 8000).cpu().numpy()
    ref = fo.spectrograms(s[None], 8000)
    assert got.shape == ref.shape and np.abs(got - ref).max() <= 2e-5 * np.abs(ref).max()


def test_generic_path_pow2_fft_vs_oracle_and_direct_dft(wav_paths, monkeypatch):
    """fft_length != 512: powers of two run the LDS radix-2 FFT kernel (O(N log N)), other lengths Bluestein's chirp-z transform on two
    power-of-two FFTs of >= min(frame, fft_length) + fft_length / 2 points (tf.signal.stft takes any fft_length, reference audio.py:229)
    -- between two LDS buffers up to 8192 points (4095 / 250 ms, 5000 / 300 ms, 12000 / 25 ms), in place in one buffer at 16 384
    (6000 / 400 ms, 10000 / 700 ms) -- and what needs more (12000 / 800 ms) the direct DFT; all against the oracle, the largest accepted
    length (16384) included, plus log-mel / MFCC on top of the FFT kernels"""
    import time
    from lidbox_amd.features import audio
    from lidbox_amd.data import tf_utils
    s, r = fo.read_wav_pcm16(wav_paths[2])
    x = _dev(np.stack([s[:16000], s[4000:20000], s[8000:24000]]))
    ref_in = np.stack([s[:16000], s[4000:20000], s[8000:24000]])
    for n_fft, len_ms in ((128, 5), (1024, 25), (4096, 100), (16384, 500), (600, 25), (1000, 60), (400, 25), (401, 25), (1009, 50), (2000, 100),
                          (4095, 250), (3, 1), (6000, 300), (5000, 300), (12000, 25), (6000, 400), (10000, 700), (12000, 800)):
        got = audio.spectrograms(x, r, frame_length_ms=len_ms, frame_step_ms=10, fft_length=n_fft).cpu().numpy()
        ref = fo.spectrograms(ref_in, r, frame_length_ms=len_ms, frame_step_ms=10, fft_length=n_fft)
        assert got.shape == ref.shape and got.shape[2] == n_fft // 2 + 1
        assert np.abs(got - ref).max() <= 2e-5 * np.abs(ref).max(), n_fft
    # the in-place kernel on plans the two-buffer kernel serves: same oracle, same bound, and the two kernels agree closely
    for n_fft, len_ms in ((600, 25), (1009, 50), (2000, 100), (3, 1), (4095, 250)):
        two = audio.spectrograms(x, r, frame_length_ms=len_ms, frame_step_ms=10, fft_length=n_fft).cpu().numpy()
        monkeypatch.setenv("LIDBOX_FEAT_BLUESTEIN_INPLACE", "1")
        one = audio.spectrograms(x, r, frame_length_ms=len_ms, frame_step_ms=10, fft_length=n_fft).cpu().numpy()
        monkeypatch.delenv("LIDBOX_FEAT_BLUESTEIN_INPLACE")
        ref = fo.spectrograms(ref_in, r, frame_length_ms=len_ms, frame_step_ms=10, fft_length=n_fft)
        assert np.abs(one - ref).max() <= 2e-5 * np.abs(ref).max(), n_fft
        assert np.abs(one - two).max() <= 2e-5 * np.abs(ref).max(), n_fft
    got = tf_utils.extract_features(x, [r] * 3, "logmelspectrogram", spec_kwargs=dict(fft_length=1024)).cpu().numpy()
    ref = fo.extract_features(ref_in, [r] * 3, "logmelspectrogram", spec_kwargs=dict(fft_length=1024))
    assert got.shape == ref.shape and np.abs(got - ref).max() < 1e-3
    got = tf_utils.extract_features(x, [r] * 3, "mfcc", spec_kwargs=dict(fft_length=2048)).cpu().numpy()
    ref = fo.extract_features(ref_in, [r] * 3, "mfcc", spec_kwargs=dict(fft_length=2048))
    assert got.shape == ref.shape and np.abs(got - ref).max() < 1e-3
    got = tf_utils.extract_features(x, [r] * 3, "logmelspectrogram", spec_kwargs=dict(fft_length=400)).cpu().numpy()      # fft_length == frame_length
    ref = fo.extract_features(ref_in, [r] * 3, "logmelspectrogram", spec_kwargs=dict(fft_length=400))
    assert got.shape == ref.shape and np.abs(got - ref).max() < 1e-3
    # the FFT kernels are not a slower way to the same numbers: 64 x 1 s at fft_length 2048 (radix-2), 2000 (Bluestein, two 4096-point
    # transforms), 2000 on the direct DFT (LIDBOX_FEAT_FORCE_DFT), and 6000 with 400-ms frames (in place, 16 384 points) against ITS direct DFT
    big = torch.randn(64, 16000, device="cuda") * 0.1
    times = {}
    for key, n_fft, len_ms, dft in (("pow2", 2048, 100, False), ("bs", 2000, 100, False), ("dft", 2000, 100, True),
                                    ("bs16k", 6000, 400, False), ("dft6000", 6000, 400, True)):
        if dft:
            monkeypatch.setenv("LIDBOX_FEAT_FORCE_DFT", "1")
        audio.spectrograms(big, 16000, frame_length_ms=len_ms, fft_length=n_fft)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            audio.spectrograms(big, 16000, frame_length_ms=len_ms, fft_length=n_fft)
        torch.cuda.synchronize()
        times[key] = (time.perf_counter() - t0) / 3
        if dft:
            monkeypatch.delenv("LIDBOX_FEAT_FORCE_DFT")
    assert times["pow2"] < times["bs"] < 0.25 * times["dft"], times
    assert times["bs16k"] < 0.25 * times["dft6000"], times


def test_linear_to_mel_standalone(wav_paths):
    from lidbox_amd.features import audio
    s, r = fo.read_wav_pcm16(wav_paths[1])
    P = fo.spectrograms(s[None], r)
    for m in range(10, 100, 15):
        got = audio.linear_to_mel(_dev(P.astype(np.float32)), r, num_mel_bins=m).cpu().numpy()
        ref = fo.linear_to_mel(P, r, num_mel_bins=m)
        assert got.shape == (1, P.shape[1], m) and not np.isnan(got).any()
        assert np.abs(got - ref).max() <= 2e-5 * np.abs(ref).max(), m


def test_pure_tone_peak_bin():
    from lidbox_amd.features import audio
    sr, k = 16000, 32
    n = np.arange(32000)
    x = np.sin(2 * np.pi * (k * sr / 512) * n / sr).astype(np.float32)[None]
    P = audio.spectrograms(_dev(x), sr).cpu().numpy()
    assert P.shape == (1, 198, 257)
    assert (P[0].argmax(axis=1) == k).all()


def test_cmvn_all_axes_and_properties():
    import lidbox_amd.features as F
    rng = np.random.default_rng(4)
    for mag in range(2, 7, 2):
        for _ in range(5):
            delta = rng.uniform(1, 10 ** mag)
            x = rng.uniform(-delta, delta, size=rng.integers(1, 20, size=3)).astype(np.float32)
            for axis in range(3):
                ym = F.cmn(_dev(x), axis=axis).cpu().numpy()
                yv = F.cmvn(_dev(x), axis=axis).cpu().numpy()
                assert not np.isnan(ym).any() and not np.isnan(yv).any()
                assert np.abs(ym.mean(axis=axis)).max() < 1
                assert np.abs(yv.mean(axis=axis)).max() < 0.1 and yv.var(axis=axis).max() < 10
                ref = fo.cmvn(x, axis=axis)
                assert np.abs(yv - ref).max() <= 1e-3 * max(1.0, np.abs(ref).max())
    assert (F.cmvn(torch.ones(2, 5, 3, device="cuda")) == 0).all()          # divide_no_nan
    x = rng.standard_normal((7, 198, 40)).astype(np.float32) * 3 + 1
    assert np.abs(F.cmvn(_dev(x)).cpu().numpy() - fo.cmvn(x)).max() <= 1e-4


def test_window_normalization():
    import lidbox_amd.features as F
    rng = np.random.default_rng(5)
    x = rng.standard_normal((3, 50, 12)).astype(np.float32)
    for w in (-1, 2, 3, 4, 7, 10, 49, 50, 300):
        for nv_ in (True, False):
            got = F.window_normalization(_dev(x), window_len=w, normalize_variance=nv_).cpu().numpy()
            ref = fo.window_normalization(x, window_len=w, normalize_variance=nv_)
            assert got.shape == x.shape and not np.isnan(got).any()
            assert np.abs(got - ref).max() <= 1e-4, (w, nv_)


def test_feature_scaling_and_power_to_db():
    import lidbox_amd.features as F
    from lidbox_amd.features import audio
    rng = np.random.default_rng(3)
    x = rng.normal(0, 50, size=(4, 30, 17)).astype(np.float32)
    y = F.feature_scaling(_dev(x), -1.0, 1.0).cpu().numpy()
    assert abs(y.min() + 1) < 1e-6 and abs(y.max() - 1) < 1e-6
    assert np.abs(y - fo.feature_scaling(x, -1.0, 1.0)).max() < 1e-5
    for axis in (0, 1, 2, -1):
        y = F.feature_scaling(_dev(x), 0.0, 5.0, axis=axis).cpu().numpy()
        assert np.abs(y.min(axis=axis)).max() < 1e-5 and np.abs(y.max(axis=axis) - 5).max() < 1e-5
        assert np.abs(y - fo.feature_scaling(x, 0.0, 5.0, axis=axis)).max() < 1e-5            # lidbox_feature_scaling_axis_fwd vs the oracle
    # a constant slice: divide_no_nan -> the lower bound; wide and tall shapes (inner > 64, R not a multiple of the row groups)
    xc = x.copy(); xc[:, :, 3] = 7.0
    y = F.feature_scaling(_dev(xc), -2.0, 2.0, axis=1).cpu().numpy()
    assert (y[:, :, 3] == -2.0).all() and np.abs(y - fo.feature_scaling(xc, -2.0, 2.0, axis=1)).max() < 1e-5
    xw = rng.normal(0, 3, size=(3, 257, 130)).astype(np.float32)
    for axis in (1, 2):
        assert np.abs(F.feature_scaling(_dev(xw), 0.0, 1.0, axis=axis).cpu().numpy() - fo.feature_scaling(xw, 0.0, 1.0, axis=axis)).max() < 1e-5
    assert F.feature_scaling(torch.zeros(0, 5, 3, device="cuda"), 0.0, 1.0, axis=1).shape == (0, 5, 3)
    # tuples of axes (tf.math.reduce_min takes them; reference features/__init__.py:7-8): adjacent and non-adjacent
    for axes in ((1, 2), (0, 1), (0, 2), (0, 1, 2), (-1, 0)):
        y = F.feature_scaling(_dev(x), -1.0, 3.0, axis=axes).cpu().numpy()
        assert y.shape == x.shape and np.abs(y - fo.feature_scaling(x, -1.0, 3.0, axis=axes)).max() < 1e-5, axes
    # more outer indices than one grid dimension holds (X[B*T, C] with axis=-1); NaN propagates like tf.reduce_min / max
    xt = rng.normal(0, 2, size=(70000, 9)).astype(np.float32)
    assert np.abs(F.feature_scaling(_dev(xt), 0.0, 1.0, axis=-1).cpu().numpy() - fo.feature_scaling(xt, 0.0, 1.0, axis=-1)).max() < 1e-5
    xn = x.copy(); xn[1, 2, 3] = np.nan
    yn = F.feature_scaling(_dev(xn), 0.0, 1.0, axis=1).cpu().numpy()
    assert np.isnan(yn[1, :, 3]).all() and not np.isnan(np.delete(yn, 3, axis=2)).any()
    assert np.isnan(F.feature_scaling(_dev(xn), 0.0, 1.0).cpu().numpy()).all()
    # audio.log10 (audio.py:162-164) on its own kernel
    p = np.abs(rng.standard_normal((5, 1000)).astype(np.float32)) + 1e-3
    got = audio.log10(_dev(p)).cpu().numpy()
    assert np.abs(got - fo.log10(p)).max() < 1e-6 and np.abs(got - np.log10(p.astype(np.float64))).max() < 1e-6
    assert torch.isneginf(audio.log10(torch.zeros(3, device="cuda"))).all() and torch.isnan(audio.log10(-torch.ones(2, device="cuda"))).all()
    S = np.abs(rng.standard_normal((2, 20, 33)).astype(np.float32)) ** 2
    for top_db in (10.0, 80.0):
        db = audio.power_to_db(_dev(S), top_db=top_db).cpu().numpy()
        assert db.max() <= 0 and db.min() >= -top_db - 1e-4
        assert np.abs(db - fo.power_to_db(S, top_db=top_db)).max() < 1e-3


def test_errors_raise():
    from lidbox_amd.data import tf_utils
    x = torch.zeros(2, 1000, device="cuda")
    with pytest.raises(ValueError):
        tf_utils.extract_features(x, [16000, 8000], "spectrogram")
    with pytest.raises(ValueError):
        tf_utils.extract_features(x[0], [16000], "spectrogram")
    with pytest.raises(Exception):
        tf_utils.extract_features(torch.zeros(2, 1000), [16000, 16000], "spectrogram")    # CPU tensor: no fallback
    with pytest.raises(FloatingPointError):
        tf_utils.extract_features(torch.full((1, 1000), float("nan"), device="cuda"), [16000], "spectrogram")


def test_steps_extract_features_config_surface(synth):
    """reference lidbox/data/steps.py:708-736 config schema: batch -> features -> unbatch, adds input + feature_type"""
    from lidbox_amd.data import steps
    sig, labels = synth
    ds = [dict(id="utt%d" % i, signal=sig[i], sample_rate=16000, target=int(labels[i])) for i in range(len(sig))]
    config = {"type": "mfcc", "spectrogram": {"frame_length_ms": 25, "frame_step_ms": 10, "fft_length": 512},
              "melspectrogram": {"num_mel_bins": 40, "fmin": 20.0, "fmax": 7000.0}, "mfcc": {"coef_begin": 1, "coef_end": 14},
              "window_normalization": {"window_len": -1, "normalize_variance": True}, "batch_size": 4}
    out = list(steps.extract_features(ds, config))
    assert [o["id"] for o in out] == [d["id"] for d in ds]
    ref = fo.extract_features(sig, [16000] * len(sig), "mfcc", melspec_kwargs=config["melspectrogram"],
                              mfcc_kwargs=config["mfcc"], window_norm_kwargs=config["window_normalization"])
    for o, r in zip(out, ref):
        assert o["feature_type"] == "mfcc" and tuple(o["input"].shape) == (198, 13)
        assert np.abs(o["input"].cpu().numpy() - r).max() <= 2e-3
    # group_by_input_length: ragged signals end up in same-length batches
    ragged = [dict(signal=sig[i][:n], sample_rate=16000) for i, n in enumerate([8000, 16000, 8000, 16000, 8000, 4000])]
    got = list(steps.extract_features(ragged, {"type": "logmelspectrogram", "group_by_input_length": {"max_batch_size": 2}}))
    assert sorted(o["input"].shape[0] for o in got) == sorted([48, 98, 48, 98, 48, 23])
    with pytest.raises(ValueError):
        list(steps.extract_features(ds, {"type": "spectrogram", "device": "/CPU"}))


@pytest.mark.parametrize("sr,mel", [(16000, dict(num_mel_bins=64)), (16000, dict(num_mel_bins=13, fmin=300.0, fmax=3400.0)),
                                    (8000, dict(num_mel_bins=40, fmax=4000.0)), (16000, dict(num_mel_bins=1)),
                                    (16000, dict(num_mel_bins=23, fmin=20.0, fmax=7600.0)),
                                    (22050, dict(num_mel_bins=40, fmax=8000.0))])
def test_fused_path_other_mel_configurations(sr, mel):
    """the segmented mel plan (runs of consecutive bins, one lane each) across band layouts: many narrow bands, few wide
    ones, a single band, band-limited and low-sample-rate filterbanks"""
    from lidbox_amd.data import tf_utils
    rng = np.random.default_rng(sr + mel["num_mel_bins"])
    sig = (rng.standard_normal((3, int(0.7 * sr))) * 0.1).astype(np.float32)
    for kind, tol in (("melspectrogram", None), ("logmelspectrogram", LOGMEL_TOL)):
        ref = fo.extract_features(sig, [sr] * 3, kind, melspec_kwargs=mel)
        got = tf_utils.extract_features(_dev(sig), [sr] * 3, kind, melspec_kwargs=mel).cpu().numpy()
        assert got.shape == ref.shape and ref.shape[2] == mel["num_mel_bins"]
        if tol is None:
            assert (np.abs(got - ref) / np.abs(ref).max()).max() <= 2e-5
        else:
            assert np.abs(got - ref).max() <= tol


def test_wide_logmel_shape_with_plans_of_growing_lds():
    """Several plans in one process through the persistent 14-wave log-mel workgroup (>= 4 tiles per CU), smallest LDS
    footprint first: the kernel's dynamic-LDS limit is raised per device, not to the first caller's size, so a later plan
    with more mel bins (more table + wave-slice bytes) still launches (features.hip LBX_LOGMEL)."""
    from lidbox_amd.data import tf_utils
    rng = np.random.default_rng(77)
    sig = (rng.standard_normal((96, 16000)) * 0.1).astype(np.float32)          # 96 x 13 tiles = 1 248 tiles
    for mel in (dict(num_mel_bins=24), dict(num_mel_bins=40), dict(num_mel_bins=64), dict(num_mel_bins=16, fmin=100.0, fmax=4000.0)):
        ref = fo.extract_features(sig[:8], [16000] * 8, "logmelspectrogram", melspec_kwargs=mel)
        got = tf_utils.extract_features(_dev(sig), [16000] * 96, "logmelspectrogram", melspec_kwargs=mel)
        torch.cuda.synchronize()
        assert got.shape == (96, 98, mel["num_mel_bins"])
        assert np.abs(got[:8].cpu().numpy() - ref).max() <= LOGMEL_TOL, mel
        # batch independence across the launch shapes: the 8-utterance call takes the 4-wave workgroups
        small = tf_utils.extract_features(_dev(sig[:8]), [16000] * 8, "logmelspectrogram", melspec_kwargs=mel)
        assert torch.equal(small, got[:8]), mel


def test_concurrent_callers_on_one_plan_are_bit_identical():
    """The reference calls the feature op from tf.data AUTOTUNE threads under one device (steps.py:734-736; SURVEY 8b
    "concurrent callers on one device").  Two host threads, each on its own HIP stream, run log-mel and MFCC through the
    SAME immutable plan over different batches, many times, while a third keeps asking for the plan: every result equals
    the single-threaded one bit for bit and no call reports an error."""
    import threading
    from lidbox_amd import _native as nv
    from lidbox_amd.features import audio
    from lidbox_amd.testutil import synthetic_batch
    plan = audio.get_plan(16000, 400, 160)
    batches = [torch.from_numpy(synthetic_batch(b, num_labels=4, duration_s=1.0, seed=b)[0]).cuda() for b in (3, 17, 64, 5)]
    kinds = (nv.FEAT_LOGMEL, nv.FEAT_MFCC)
    expect = {(i, k): plan.run(k, x).clone() for i, x in enumerate(batches) for k in kinds}
    torch.cuda.synchronize()
    errors, results = [], {}

    def worker(tid):
        try:
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                for rep in range(25):
                    for i, x in enumerate(batches):
                        if (i + tid) % 2:
                            continue
                        p = audio.get_plan(16000, 400, 160)              # the shared, cached plan
                        assert p is plan
                        for k in kinds:
                            out = p.run(k, x)
                            if rep == 24:
                                results[(i, k)] = out
            stream.synchronize()
        except Exception as e:                                          # noqa: BLE001 - surfaced below
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert set(results) == set(expect)
    for key, ref in expect.items():
        assert torch.equal(results[key], ref), key


@pytest.mark.parametrize("B,dur", [(6, 2.0), (300, 1.0), (3, 0.37)])
def test_bf16_shadow_equals_rounded_features(B, dur):
    """lidbox_extract_features_fwd_shadow: out16 == bf16(out) bit for bit (round-to-nearest-even), and `out` is what the call
    without a shadow writes -- the log-mel kernel's own shadow store (4-wave and wide workgroup shapes), the conversion
    pass of the other kinds, the generic (non-512 FFT) path, and a padded batch stride whose gap stays untouched."""
    from lidbox_amd import _native as nv
    from lidbox_amd.features import audio
    from lidbox_amd.testutil import synthetic_batch
    x = torch.from_numpy(synthetic_batch(B, num_labels=4, duration_s=dur, seed=B)[0]).cuda()
    plans = [audio.get_plan(16000, 400, 160)] + ([audio.get_plan(16000, 400, 160, fft_length=1024)] if B == 6 else [])
    for plan in plans:
        for kind in (nv.FEAT_LOGMEL, nv.FEAT_MEL, nv.FEAT_MFCC, nv.FEAT_SPECTROGRAM):
            ref = plan.run(kind, x)
            out16 = torch.full(ref.shape, 7.0, dtype=torch.bfloat16, device="cuda")
            got = plan.run(kind, x, out16=out16)
            assert torch.equal(got, ref)
            assert torch.equal(out16, ref.to(torch.bfloat16)), (kind, plan.fft_length)
    # strided: utterances 3 rows apart more than dense; the gap rows keep their fill in both buffers
    plan = plans[0]
    T, C = plan.num_frames(x.shape[1]), 40
    buf = torch.full((B, T + 3, C), -5.0, device="cuda")
    buf16 = torch.full((B, T + 3, C), -5.0, dtype=torch.bfloat16, device="cuda")
    plan.run(nv.FEAT_LOGMEL, x, out=buf, out_batch_stride=(T + 3) * C, out16=buf16)
    ref = plan.run(nv.FEAT_LOGMEL, x)
    assert torch.equal(buf[:, :T], ref) and torch.equal(buf16[:, :T], ref.to(torch.bfloat16))
    assert (buf[:, T:] == -5.0).all() and (buf16[:, T:] == -5.0).all()


# ------------------------------------------------------------------ round 6: the streaming kernel's extras
@pytest.mark.parametrize("kind", ["logmelspectrogram", "mfcc", "melspectrogram", "spectrogram"])
def test_pcm16_source_read_in_place_is_bit_identical_to_convert_then_extract(kind, wav_paths):
    """16-bit PCM handed to the feature kernel as it lies in memory (reference audio.py:17-23 read_wav -> tf.audio.decode_wav: value /
    32768) against lidbox_pcm16_to_f32 followed by the float call: same bits (the scale is a power of two folded into the window
    table), on random PCM with full-scale values, on the reference's WAV fixtures, and against the float64 oracle."""
    from lidbox_amd.data import tf_utils
    from lidbox_amd.features import audio
    rng = np.random.default_rng(5)
    pcm = rng.integers(-32768, 32768, size=(37, 16000), dtype=np.int16)
    pcm[0, :8] = [-32768, 32767, 0, 1, -1, 255, -256, 12345]
    rates = [16000] * len(pcm)
    x16 = _dev(pcm)
    got = tf_utils.extract_features(x16, rates, kind)
    xf = audio.pcm16_to_float(x16.reshape(-1), 1).reshape(pcm.shape)
    assert torch.equal(xf.cpu(), torch.from_numpy(pcm.astype(np.float32) / 32768.0))
    want = tf_utils.extract_features(xf, rates, kind)
    assert torch.equal(got, want)
    ref = fo.extract_features(pcm[:4].astype(np.float64) / 32768.0, rates[:4], kind)
    if kind in ("logmelspectrogram", "mfcc"):
        assert np.abs(got[:4].cpu().numpy() - ref).max() <= LOGMEL_TOL
    else:
        assert np.abs(got[:4].cpu().numpy() - ref).max() <= 2e-5 * np.abs(ref).max()
    # the reference's fixtures, one utterance per call (odd lengths: the tail of the last float4 is outside the utterance)
    for path in wav_paths:
        sig64, rate = fo.read_wav_pcm16(path)
        raw = np.round(sig64 * 32768.0).astype(np.int16)
        a = tf_utils.extract_features(_dev(raw[None, :]), [rate], kind)
        b = tf_utils.extract_features(_dev((raw.astype(np.float32) / 32768.0)[None, :]), [rate], kind)
        assert torch.equal(a, b), path
    # a source the in-place kernel cannot take (rows 2 bytes off an 8-byte boundary) goes through one conversion pass: same result
    big = _dev(np.concatenate([np.zeros((37, 1), np.int16), pcm], axis=1))
    assert torch.equal(tf_utils.extract_features(big[:, 1:], rates, kind), want)


def test_nonfinite_flag_from_the_store_stage():
    """tf.debugging.assert_all_finite of reference tf_utils.py:173-194 without a pass over the output: the kernels fold "a value
    I stored is not finite" into a 4-byte flag.  Clean signals leave it 0; one NaN / Inf sample anywhere inside a frame sets it and
    extract_features raises like the reference; a poisoned sample that no frame reads (the tail behind the last frame) does not."""
    from lidbox_amd import _native as nv
    from lidbox_amd.data import tf_utils
    from lidbox_amd.features import audio
    rng = np.random.default_rng(9)
    sig = (rng.standard_normal((300, 16000 + 77)) * 0.1).astype(np.float32)          # 300 x 13 tiles: the streaming kernel, several tiles per wave
    plan = audio.get_plan(16000, 400, 160)
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    for kind in (nv.FEAT_LOGMEL, nv.FEAT_MFCC, nv.FEAT_MEL, nv.FEAT_SPECTROGRAM):
        x = _dev(sig)
        flag.zero_()
        out = plan.run(kind, x, nonfinite=flag)
        assert audio.all_finite(flag) and bool(torch.isfinite(out).all())
        for b, n, v in ((0, 0, np.nan), (299, 15900, np.inf), (150, 8000, -np.inf)):
            x = _dev(sig)
            x[b, n] = v
            flag.zero_()
            out = plan.run(kind, x, nonfinite=flag)
            assert not audio.all_finite(flag), (kind, b, n)
            assert not bool(torch.isfinite(out).all())
        x = _dev(sig)
        x[7, 16000 + 50] = np.nan          # behind the last frame (98 frames: the last one covers samples 15 520 ... 15 919)
        flag.zero_()
        out = plan.run(kind, x, nonfinite=flag)
        assert audio.all_finite(flag) and bool(torch.isfinite(out).all())
    x = _dev(sig)
    x[3, 1234] = np.nan
    with pytest.raises(FloatingPointError):
        tf_utils.extract_features(x, [16000] * 300, "logmelspectrogram")
    assert tf_utils.extract_features(x, [16000] * 300, "logmelspectrogram", check_finite=False).shape == (300, 98, 40)
    # the paths without a flag in their store stage (another fft_length: generic kernels) report through one extra pass
    plan1k = audio.get_plan(16000, 400, 160, fft_length=1024)
    flag.zero_()
    plan1k.run(nv.FEAT_LOGMEL, x[:4], nonfinite=flag)
    assert not audio.all_finite(flag)
    flag.zero_()
    plan1k.run(nv.FEAT_LOGMEL, _dev(sig[:4]), nonfinite=flag)
    assert audio.all_finite(flag)


@pytest.mark.parametrize("frame_ms,step_ms", [(32, 10), (25, 10), (26, 8), (20, 5)])
def test_streaming_kernel_frame_lengths_and_the_unaligned_kernel_agree(frame_ms, step_ms):
    """Frames of <= 416 samples take the pruned instantiation (13 of 16 sample loads per lane), longer ones (32 ms = 512 samples) the
    full one; signals whose rows are not 16-byte aligned take the round-1 kernel with its guarded loads.  All against the oracle,
    and the two kernels against each other (same arithmetic, different instruction selection: within 2e-5 of the largest value)."""
    from lidbox_amd.data import tf_utils
    rng = np.random.default_rng(frame_ms)
    B, N = 130, 12000 + 3                                                      # 130 utterances: several tiles per wave
    sig = (rng.standard_normal((B, N + 1)) * 0.1).astype(np.float32)
    kw = dict(spec_kwargs=dict(frame_length_ms=frame_ms, frame_step_ms=step_ms))
    dev = _dev(sig)
    for kind, tol in (("logmelspectrogram", LOGMEL_TOL), ("mfcc", MFCC_TOL)):
        ref = fo.extract_features(sig[:3, :N], [16000] * 3, kind, **kw)
        aligned = tf_utils.extract_features(dev[:, :N], [16000] * B, kind, **kw)              # rows 48 016 bytes apart: 16-byte aligned
        unaligned = tf_utils.extract_features(dev[:, 1:N + 1], [16000] * B, kind, **kw)      # 4 bytes off
        ref_u = fo.extract_features(sig[:3, 1:N + 1], [16000] * 3, kind, **kw)
        assert aligned.shape[1:] == ref.shape[1:]
        assert np.abs(aligned[:3].cpu().numpy() - ref).max() <= tol
        assert np.abs(unaligned[:3].cpu().numpy() - ref_u).max() <= tol
    spec_a = tf_utils.extract_features(dev[:, :N], [16000] * B, "spectrogram", **kw)
    spec_u = tf_utils.extract_features(dev[:, 4:N + 4 - 3].contiguous(), [16000] * B, "spectrogram", **kw)      # same kernel, shifted copy
    assert spec_a.shape[0] == spec_u.shape[0] == B
    # one input through both kernels: a dense copy (aligned) vs a view one float off (unaligned) of the same samples
    same_a = tf_utils.extract_features(dev[:, 1:N + 1].contiguous(), [16000] * B, "spectrogram", **kw)
    same_u = tf_utils.extract_features(dev[:, 1:N + 1], [16000] * B, "spectrogram", **kw)
    assert float((same_a - same_u).abs().max()) <= 2e-5 * float(same_a.abs().max())
