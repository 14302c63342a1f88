"""
Pins oracle/ (the CPU restatement) against everything the reference's own tests
hold for this path, the closed-form known
answers derived from the reference's demo data, and independent libraries.

Reference tests restated here (file:line relative to /root/reference/):
  tests/test_features_audio.py:125-129  test_ms_to_frames      (exact)
  tests/test_features_audio.py:131-145  test_spectrograms      (shape law, no NaN)
  tests/test_features_audio.py:147-155  test_linear_to_mel     (shape, no NaN)
  tests/test_features_audio.py:99-104   test_fft_frequencies
  tests/test_features_audio.py:106-113  test_log10             (1e-6)
  tests/test_features_audio.py:115-123  test_power_to_db       (max <= 0)
  tests/test_features.py:14-26          test_feature_scaling   (1e-9)
  tests/test_features.py:28-43          test_cmvn
  tests/test_features.py:45-58          test_window_normalization
  tests/test_models.py:104-107,65-68    test_xvector / test_cnn (shape, no NaN, T=1)
  lidbox/losses.py:55-100, lidbox/metrics.py:122-164  self-demos (known answers)
"""
import numpy as np
import pytest
import scipy.fft
import torch

from oracle import features_np as fo
from oracle import model_np as mo
from oracle import torch_ref as tr


# ------------------------------------------------------------------ features
def test_ms_to_frames_reference_grid():
    for sr in range(1000, 60000, 1000):
        for ms in range(1, 5000, 100):
            assert fo.ms_to_frames(sr, ms) == (sr // 1000) * ms
    assert fo.ms_to_frames(16000, 25) == 400 and fo.ms_to_frames(16000, 10) == 160


def test_hann_window_periodic_and_odd():
    w = fo.hann_window(400)
    n = np.arange(400)
    assert np.allclose(w, 0.5 - 0.5 * np.cos(2 * np.pi * n / 400), atol=1e-15)
    assert w[0] == 0 and abs(w[200] - 1) < 1e-15
    # odd length: TF's "periodic" formula degenerates to the symmetric window
    w = fo.hann_window(401)
    assert np.allclose(w, np.hanning(401), atol=1e-12)
    assert fo.hann_window(1)[0] == 1.0


def test_spectrograms_shape_law(wav_paths):
    for path in wav_paths:
        s, r = fo.read_wav_pcm16(path)
        assert s.shape == (48000,) and r == 16000 and not np.isnan(s).any()
        for len_ms in range(20, 101, 20):
            for n_fft in (256, 512, 1024, 2048):
                if n_fft < fo.ms_to_frames(r, len_ms):
                    continue
                step_ms = len_ms // 2
                P = fo.spectrograms(s[None], r, frame_length_ms=len_ms, frame_step_ms=step_ms,
                                    fft_length=n_fft)[0]
                assert not np.isnan(P).any()
                assert P.shape[0] == s.shape[0] // fo.ms_to_frames(r, step_ms) - 1
                assert P.shape[1] == n_fft // 2 + 1


def test_stft_matches_direct_dft():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 1000))
    S = fo.stft(x, 400, 160, 512)
    assert S.shape == (2, 4, 257)
    # direct O(N^2) DFT of frame 2 of signal 1, right-zero-padded to 512
    fr = np.zeros(512)
    fr[:400] = x[1, 320:720] * fo.hann_window(400)
    k = np.arange(257)[:, None]
    n = np.arange(512)[None, :]
    D = (fr[None, :] * np.exp(-2j * np.pi * k * n / 512)).sum(axis=1)
    assert np.abs(S[1, 2] - D).max() < 1e-10


def test_pure_tone_peaks_at_bin_centre():
    sr, k = 16000, 32
    n = np.arange(32000)
    x = np.sin(2 * np.pi * (k * sr / 512) * n / sr)[None]
    P = fo.spectrograms(x, sr)
    assert P.shape == (1, 198, 257)
    assert (P[0].argmax(axis=1) == k).all()


def test_mel_matrix_lidbox_quirk():
    W = fo.linear_to_mel_weight_matrix(40, 257, 16000, 0.0, 8000.0)
    assert W.shape == (257, 40)
    assert (W[0] == 0).all()                      # DC row zero (HTK)
    assert np.count_nonzero(W) == 464
    nz = (W > 0).sum(axis=0)
    assert nz.min() == 2 and nz.max() == 30
    assert ((W > 0).sum(axis=1) <= 2).all()       # every bin feeds at most two bands
    # non-endpoint linspace: last FFT bin sits at nyq*256/257, top band edge below fmax
    edges = fo._linspace(fo._hertz_to_mel(0.0, np.float64), fo._hertz_to_mel(8000.0, np.float64), 42,
                         np.float64)
    top_hz = 700.0 * (np.exp(edges[-1] / 1127.0) - 1.0)
    assert 7480 < top_hz < 7500
    # it is NOT the stock TF matrix (endpoint linspace): differs by ~0.9 somewhere
    lin = np.linspace(0, 8000.0, 257)[1:]
    e2 = np.linspace(fo._hertz_to_mel(0.0, np.float64), fo._hertz_to_mel(8000.0, np.float64), 42)
    sm = fo._hertz_to_mel(lin, np.float64)[:, None]
    stock = np.maximum(0, np.minimum((sm - e2[:40]) / (e2[1:41] - e2[:40]),
                                     (e2[2:42] - sm) / (e2[2:42] - e2[1:41])))
    assert np.abs(W[1:] - stock).max() > 0.5
    # float32 op-order build agrees with the float64 one
    W32 = fo.linear_to_mel_weight_matrix(40, 257, 16000, 0.0, 8000.0, np.float32)
    assert np.abs(W32 - W).max() < 2e-4


def test_mel_matrix_body_matches_independent_htk_filterbank():
    """Third-party pin of the filterbank formulas (hertz->mel, edges, slopes, max/min, DC row): with the endpoint
    linspace of the TensorFlow file that mel_ops.py:1-6 says it was copied from, the oracle's matrix must equal the
    HTK filterbank of an unrelated implementation (transformers.audio_utils.mel_filter_bank, triangles in mel space,
    no area normalisation).  The reference's own matrix then differs from that one ONLY through its vendored
    non-endpoint _linspace (mel_ops.py:11-16), which the previous test pins."""
    au = pytest.importorskip("transformers.audio_utils")
    for M, F, sr, lo, hi in ((40, 257, 16000, 0.0, 8000.0), (20, 129, 8000, 125.0, 3800.0), (64, 513, 22050, 20.0, 11025.0),
                             (13, 257, 16000, 300.0, 3400.0)):
        W = fo.linear_to_mel_weight_matrix(M, F, sr, lo, hi, np.float64, stock_linspace=True)
        ref = au.mel_filter_bank(F, M, lo, hi, sr, norm=None, mel_scale="htk", triangularize_in_mel_space=True)
        assert W.shape == ref.shape == (F, M)
        assert np.abs(W[1:] - ref[1:]).max() < 1e-9            # row 0 (DC) is zeroed by TF's HTK convention (mel_ops.py:37,74)
        assert np.all(W[0] == 0)


def test_power_spectrogram_matches_independent_stft(wav_paths):
    """|STFT|^2 of the oracle (frame, periodic Hann, right zero-pad to 512, rfft: audio.py:219-230) against the
    un-centred power spectrogram of transformers.audio_utils (an unrelated framing / window / FFT code path)."""
    au = pytest.importorskip("transformers.audio_utils")
    win = au.window_function(400, "hann", periodic=True)
    assert np.abs(win - fo.hann_window(400, True, np.float64)).max() < 1e-12
    for path in wav_paths[:3]:
        s, r = fo.read_wav_pcm16(path)
        s = s[:r].astype(np.float64)
        P = fo.spectrograms(s[None], r)[0]
        ref = au.spectrogram(s, win, 400, 160, fft_length=512, power=2.0, center=False, dtype=np.float64).T
        assert P.shape == ref.shape
        assert np.abs(P - ref).max() <= 1e-6 * np.abs(ref).max()        # that library transforms in complex64


def test_linear_to_mel_shapes(wav_paths):
    for path in wav_paths:
        s, r = fo.read_wav_pcm16(path)
        P = fo.spectrograms(s[None], r)
        for m in range(10, 100, 15):
            mel = fo.linear_to_mel(P, r, num_mel_bins=m)[0]
            assert not np.isnan(mel).any()
            assert mel.shape == (P.shape[1], m)


def test_fft_frequencies():
    for sr in range(4000, 60000, 4000):
        for n_fft in (2 ** i for i in range(1, 13)):
            a = fo.fft_frequencies(sr, n_fft)
            b = np.fft.rfftfreq(n_fft, 1.0 / sr)        # == librosa.fft_frequencies
            b[-1] = abs(b[-1])
            if sr % 2 == 0:
                assert np.abs(a - b).max() < 1e-9


def test_log10():
    rng = np.random.default_rng(1)
    for rank in range(1, 5):
        for _ in range(5):
            x = np.maximum(1e-12, rng.normal(1e6, 1e4, size=rng.integers(1, 10, size=rank)))
            assert np.abs(np.log10(x) - fo.log10(x.astype(np.float32), np.float32)).max() < 1e-6


def test_power_to_db_bound(wav_paths):
    import scipy.signal
    for top_db in range(10, 110, 10):
        for path in wav_paths:
            s, r = fo.read_wav_pcm16(path)
            _, _, st = scipy.signal.stft(s)
            db = fo.power_to_db((np.abs(st) ** 2)[None], top_db=float(top_db))[0]
            assert not np.isnan(db).any() and db.max() <= 0
            assert db.min() >= -top_db - 1e-9


def test_mfcc_matches_scipy_dct():
    rng = np.random.default_rng(2)
    X = rng.standard_normal((3, 7, 40))
    ref = scipy.fft.dct(X, type=2, axis=-1) / np.sqrt(2 * 40)
    assert np.abs(fo.mfccs_from_log_mel(X, 0, 40) - ref).max() < 1e-12
    assert fo.mfccs_from_log_mel(X).shape == (3, 7, 12)
    assert np.abs(fo.mfccs_from_log_mel(X) - ref[..., 1:13]).max() < 1e-12


def test_feature_scaling():
    rng = np.random.default_rng(3)
    for rank in range(1, 5):
        for _ in range(30):
            delta = rng.uniform(1, 1e3)
            mn = rng.uniform(-delta, delta)
            mx = mn + rng.uniform(0, delta / 2)
            x = rng.normal(0, delta ** 2, size=rng.integers(2, 20, size=rank))
            for axis in [None] + list(range(rank)):
                y = fo.feature_scaling(x, mn, mx, axis=axis)
                assert not np.isnan(y).any() and y.shape == x.shape
                assert np.abs(y.min(axis=axis) - mn).max() < 1e-9
                assert np.abs(y.max(axis=axis) - mx).max() < 1e-9


def test_cmvn():
    rng = np.random.default_rng(4)
    for mag in range(2, 7):
        for _ in range(20):
            delta = rng.uniform(1, 10 ** mag)
            x = rng.uniform(-delta, delta, size=rng.integers(1, 20, size=3))
            for axis in range(3):
                ym = fo.cmn(x, axis=axis)
                assert not np.isnan(ym).any() and ym.shape == x.shape
                assert np.abs(ym.mean(axis=axis)).max() < 1
                yv = fo.cmvn(x, axis=axis)
                assert not np.isnan(yv).any() and yv.shape == x.shape
                assert np.abs(yv.mean(axis=axis)).max() < 0.1
                assert yv.var(axis=axis).max() < 10
    # constant channel -> 0 (divide_no_nan), not NaN
    assert (fo.cmvn(np.ones((2, 5, 3))) == 0).all()


def test_window_normalization():
    rng = np.random.default_rng(5)
    for _ in range(20):
        delta = rng.uniform(1, 1e3)
        x = rng.uniform(-delta, delta, size=rng.integers(1, 20, size=3))
        for w in [-1] + list(range(2, x.shape[0] + 1)):
            for nv in (True, False):
                y = fo.window_normalization(x, axis=1, window_len=w, normalize_variance=nv)
                assert not np.isnan(y).any() and y.shape == x.shape
    # sliding branch against a direct per-frame loop with numpy 'reflect'
    x = rng.standard_normal((2, 11, 3))
    for w in (2, 3, 4, 5, 10):
        y = fo.window_normalization(x, window_len=w)
        xp = np.pad(x, [(0, 0), (w // 2, w // 2 - 1 + (w & 1)), (0, 0)], mode="reflect")
        for t in range(11):
            win = xp[:, t:t + w]
            assert np.allclose(y[:, t], (x[:, t] - win.mean(1)) / win.std(1), atol=1e-12)


def test_extract_features_dispatch():
    from lidbox_amd.testutil import synthetic_batch
    sig, _ = synthetic_batch(3)
    sr = [16000] * 3
    assert fo.extract_features(sig, sr, "spectrogram").shape == (3, 198, 257)
    assert fo.extract_features(sig, sr, "melspectrogram").shape == (3, 198, 40)
    lm = fo.extract_features(sig, sr, "logmelspectrogram")
    assert lm.shape == (3, 198, 40) and np.isfinite(lm).all()
    mf = fo.extract_features(sig, sr, "mfcc", window_norm_kwargs=dict(window_len=-1))
    assert mf.shape == (3, 198, 12)
    assert np.abs(mf.mean(axis=1)).max() < 1e-9
    assert fo.extract_features(sig, sr, "db_spectrogram").max() <= 0
    with pytest.raises(ValueError):
        fo.extract_features(sig, [16000, 8000, 16000], "spectrogram")
    with pytest.raises(ValueError):
        fo.extract_features(sig[0], sr, "spectrogram")
    # float32 op-order path vs float64 truth
    lm32 = fo.extract_features(sig, sr, "logmelspectrogram", dtype=np.float32)
    assert np.abs(lm32 - lm).max() < 1e-3
    # torch-CPU restatement (the timed cpu_baseline) agrees too
    lmt = tr.LogMelCPU()(torch.from_numpy(sig)).numpy()
    assert np.abs(lmt - lm).max() < 1e-3


# ------------------------------------------------------------------ models
def test_conv_shapes_and_causality():
    x = np.random.default_rng(6).standard_normal((2, 198, 40)).astype(np.float32)
    p = mo.xvector_init(40, 4)
    lens = []
    h = x
    for name, f, k, s in mo.XVECTOR_FRAMES:
        h = mo.conv1d_causal_fwd(h, p[name + ".W"], p[name + ".b"], s)
        lens.append(h.shape[1])
    assert lens == [198, 99, 33, 33, 33]
    n_params = sum(v.size for v in p.values())
    assert n_params == 4510176
    assert sum(v.size for v in mo.xvector_init(40, 100).values()) == 4559424
    # causal: output t must not depend on inputs after t*s
    W, b = p["frame2.W"], p["frame2.b"]
    a = np.random.default_rng(7).standard_normal((1, 20, 512)).astype(np.float32)
    a2 = a.copy()
    a2[:, 11:] += 1.0
    y1, y2 = mo.conv1d_causal_fwd(a, W, b, 2), mo.conv1d_causal_fwd(a2, W, b, 2)
    assert np.array_equal(y1[:, :6], y2[:, :6]) and not np.array_equal(y1[:, 6:], y2[:, 6:])


@pytest.mark.parametrize("shape", [(1, 1, 1), (3, 1, 7), (2, 5, 40), (4, 50, 13), (2, 198, 40)])
def test_xvector_and_cnn_valid_output(shape):
    rng = np.random.default_rng(8)
    x = rng.uniform(-1e3, 1e3, size=shape).astype(np.float32)
    for n_out in (1, 4, 100):
        y = mo.xvector_fwd(mo.xvector_init(shape[2], n_out), x)
        assert y.shape == (shape[0], n_out) and not np.isnan(y).any()
        e = mo.xvector_fwd(mo.xvector_init(shape[2], n_out), x, embedding=True)
        assert e.shape == (shape[0], 512)
        y = mo.cnn_fwd(mo.cnn_init(shape[2], n_out), x)
        assert y.shape == (shape[0], n_out) and not np.isnan(y).any()


def test_stats_pool_T1_and_clip():
    x = np.random.default_rng(9).standard_normal((2, 1, 5))
    out = mo.stats_pool_fwd(x)
    assert np.allclose(out[:, :5], x[:, 0]) and np.allclose(out[:, 5:], 1e-5)


def test_forward_matches_torch_conv1d():
    rng = np.random.default_rng(10)
    x = rng.standard_normal((3, 50, 24)).astype(np.float32)
    p = mo.xvector_init(24, 7, seed=3)
    y_np = mo.xvector_fwd({k: v.astype(np.float64) for k, v in p.items()}, x.astype(np.float64))
    y_t = tr.xvector_fwd(tr.to_torch_params(p, False, torch.float64), torch.tensor(x, dtype=torch.float64))
    assert np.abs(y_np - y_t.numpy()).max() < 1e-10
    pc = mo.cnn_init(24, 7, seed=4)
    y_np = mo.cnn_fwd({k: v.astype(np.float64) for k, v in pc.items()}, x.astype(np.float64))
    y_t = tr.cnn_fwd(tr.to_torch_params(pc, False, torch.float64), torch.tensor(x, dtype=torch.float64))
    assert np.abs(y_np - y_t.numpy()).max() < 1e-10


def test_backward_matches_torch_autograd():
    rng = np.random.default_rng(11)
    x = rng.standard_normal((4, 37, 16))
    y = rng.integers(0, 5, size=4)
    p = {k: v.astype(np.float64) for k, v in mo.xvector_init(16, 5, seed=5).items()}
    # non-zero biases so the bias path is exercised
    for k in p:
        if k.endswith(".b"):
            p[k] = rng.standard_normal(p[k].shape) * 0.1
    loss, g, _ = mo.xvector_loss_and_grads(p, x, y)
    pt = tr.to_torch_params(p, True, torch.float64)
    lt = tr.sparse_ce_from_logits(tr.xvector_fwd(pt, torch.tensor(x)), torch.tensor(y))
    lt.backward()
    assert abs(loss - float(lt.detach())) < 1e-12
    for k in p:
        ref = pt[k].grad.numpy()
        assert np.abs(g[k] - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max()), k


def test_adam_matches_torch_ref_and_closed_form():
    rng = np.random.default_rng(12)
    p = {"w": rng.standard_normal(10)}
    m = {"w": np.zeros(10)}
    v = {"w": np.zeros(10)}
    pt = {"w": torch.tensor(p["w"].copy(), requires_grad=True)}
    opt = tr.KerasAdam(pt)
    w0 = p["w"].copy()
    for t in range(1, 4):
        g = rng.standard_normal(10)
        pt["w"].grad = torch.tensor(g)
        mo.adam_step(p, {"w": g}, m, v, t)
        opt.step()
        if t == 1:   # closed form of the first Keras-Adam step: lr * g / (|g| + eps/sqrt(1-b2))
            assert np.allclose(p["w"], w0 - 1e-3 * g / (np.abs(g) + 1e-7 / np.sqrt(1 - 0.999)), atol=1e-12)
        assert np.abs(p["w"] - pt["w"].detach().numpy()).max() < 1e-15


# ------------------------------------------------------------------ AP loss / C_avg known answers
def _sigmoid(x):
    return 1 / (1 + np.exp(-x))


def test_ap_loss_demo_known_answers():
    """lidbox/losses.py:61-97: N=3, D=100, one-hot language vectors (noise std 0)."""
    N, D = 3, 100
    y_true = np.array([0, 1, 1, 1, 0, 2, 1, 2], np.int32)
    cases = [[0, 1, 1, 1, 0, 2, 1, 2], [0, 1, 1, 2, 0, 2, 1, 2], [1, 2, 0, 2, 1, 1, 0, 1]]
    expect = []
    for pred in cases:
        z = mo.l2_normalize(np.eye(D)[pred])
        expect.append(mo.ap_loss(y_true, z, N))
    correct = (N - 1) * _sigmoid(-np.pi / 2)             # 0.3442058
    wrong = _sigmoid(np.pi / 2) + (N - 2) / 2            # 1.3278971
    assert abs(correct - 0.3442058) < 1e-6 and abs(wrong - 1.3278971) < 1e-6
    assert abs(expect[0] - correct) < 1e-12
    assert abs(expect[1] - (7 * correct + wrong) / 8) < 1e-12 and abs(expect[1] - 0.4671672) < 1e-6
    assert abs(expect[2] - wrong) < 1e-12
    # torch restatement agrees
    z = mo.l2_normalize(np.eye(D)[cases[1]])
    assert abs(float(tr.ap_loss(torch.tensor(y_true.astype(np.int64)), torch.tensor(z), N)) - expect[1]) < 1e-12


def test_ap_loss_grad_matches_autograd():
    rng = np.random.default_rng(13)
    z = mo.l2_normalize(rng.standard_normal((6, 12)))
    y = rng.integers(0, 5, size=6)
    g = mo.ap_loss_grad(y, z, 5, delta_weight=1.7)
    zt = torch.tensor(z, requires_grad=True)
    tr.ap_loss(torch.tensor(y), zt, 5, 1.7).backward()
    assert np.abs(g - zt.grad.numpy()).max() < 1e-12


def test_cavg_demo_known_answers():
    """lidbox/metrics.py:127-164."""
    true_pos = np.array([[1, 0, 0], [0, 1, 0], [0, 1, 0], [0, 1, 0], [1, 0, 0], [0, 0, 1], [0, 1, 0],
                         [0, 0, 1]], np.float32)
    with np.errstate(divide="ignore"):
        pred = np.log(np.array([[.1, .2, .9], [.9, .2, .0], [.1, .9, .0], [.2, .8, .5], [.6, .3, .1],
                                [.1, .0, .7], [.1, .0, .7], [.9, .1, .0]], np.float32))
    th = np.log(np.array([0.05, 0.4, 0.6, 0.95], np.float32))
    c = mo.AverageDetectionCost(3, th)
    c.update_state(true_pos, pred)
    P_miss, P_fa, C = c.per_threshold()
    assert np.allclose(P_miss, [0.25, 0.5, 0.5, 1.0])
    assert np.allclose(P_fa, [0.8333333, 0.2916667, 0.25, 0.0], atol=1e-6)
    assert np.allclose(C, [0.5416667, 0.3958333, 0.375, 0.5], atol=1e-6)
    assert abs(c.result() - 0.375) < 1e-7
    # l == m pair counters stay zero (metrics.py:105-111)
    for l in range(3):
        assert c.fp_pairs[l, l].sum() == 0 and c.tn_pairs[l, l].sum() == 0
    c.reset_states()
    assert c.result() == 0.0                                  # metrics.py:163-164
    s = mo.SparseAverageDetectionCost(3, th)
    s.update_state(true_pos.argmax(1), pred)
    assert abs(s.result() - 0.375) < 1e-7


# ------------------------------------------------------------------ SURVEY 8f.1 variants
@pytest.mark.parametrize("d", [1, 2, 3])
def test_oracle_dilated_conv_matches_torch_autograd(d):
    """Keras Conv1D(padding="causal", dilation_rate=d) restated in model_np vs torch conv1d + autograd"""
    import torch
    import torch.nn.functional as F
    from oracle import model_np as mo
    rng = np.random.default_rng(d)
    x, W, b = rng.standard_normal((2, 17, 5)), rng.standard_normal((3, 5, 4)) * 0.3, rng.standard_normal(4)
    y = mo.conv1d_causal_fwd(x, W, b, 1, relu=True, d=d)
    xt, Wt, bt = (torch.tensor(v, requires_grad=True) for v in (x, W, b))
    yt = torch.relu(F.conv1d(F.pad(xt.transpose(1, 2), (2 * d, 0)), Wt.permute(2, 1, 0), bt, dilation=d)).transpose(1, 2)
    assert np.abs(y - yt.detach().numpy()).max() < 1e-12
    dy = rng.standard_normal(y.shape)
    yt.backward(torch.tensor(dy))
    dx, dW, db = mo.conv1d_causal_bwd(x, W, y, dy, 1, d=d)
    for got, ref in ((dx, xt.grad), (dW, Wt.grad), (db, bt.grad)):
        assert np.abs(got - ref.numpy()).max() < 1e-12


def test_oracle_frequency_attention_matches_torch_autograd():
    """clstm.py:31-42 restated in model_np.freq_attention_fwd/_bwd vs a torch transcription + autograd"""
    import torch
    from oracle import model_np as mo
    rng = np.random.default_rng(5)
    H = np.abs(rng.standard_normal((2, 7, 12)))
    W1, W2 = rng.standard_normal((12, 5)) * 0.4, rng.standard_normal((5, 4)) * 0.4
    Hw, cache = mo.freq_attention_fwd(H, W1, W2, True)
    Ht, W1t, W2t = (torch.tensor(v, requires_grad=True) for v in (H, W1, W2))
    FA = torch.softmax(torch.relu(Ht @ W1t) @ W2t, -1)
    Hwt = (Ht.reshape(2, 7, 4, 3) * FA[..., None]).reshape(2, 7, 12)
    assert np.abs(Hw - Hwt.detach().numpy()).max() < 1e-12
    assert np.allclose(cache["F"].sum(-1), 1.0)
    d = rng.standard_normal(Hw.shape)
    Hwt.backward(torch.tensor(d))
    for got, ref in zip(mo.freq_attention_bwd(H, W1, W2, cache, d), (Ht.grad, W1t.grad, W2t.grad)):
        assert np.abs(got - ref.numpy()).max() < 1e-12
    with pytest.raises(AssertionError):                     # clstm.py:32: channels must divide into the bins
        mo.freq_attention_fwd(H, W1, rng.standard_normal((5, 5)))


def test_xvector_2d_oracle_numpy_equals_torch_and_batchnorm_properties():
    """8f.1: the numpy restatement of xvector_2d (Conv2D along frequency + BatchNormalization) against an independent torch
    formulation (conv1d over the frequency axis of every frame), training and inference statistics; BatchNormalization
    properties: normalised training output has zero mean / unit variance (up to epsilon) per channel, the running mean
    moves by (1 - momentum) towards the batch mean, the running variance towards the Bessel-corrected batch variance (the
    fused kernel's estimate tf.keras keeps for the running average; numpy: var(ddof=1))."""
    import torch
    from oracle import torch_ref as tr
    rng = np.random.default_rng(21)
    p = mo.xvector_2d_init(40, 5, seed=3, dtype=np.float64)
    assert mo.xvector_2d_freq_dims(40) == [40, 36, 17, 5, 1] and p["frame1.W"].shape == (5, 32, 512)
    assert p["frame2d_2_conv.W"].shape == (1, 3, 256, 128) and p["frame2d_4_bn.moving_variance"].shape == (32,)
    for k in p:
        if k.endswith("gamma"): p[k] = rng.uniform(0.5, 1.5, p[k].shape)
        if k.endswith("beta") or k.endswith("_conv.b"): p[k] = rng.standard_normal(p[k].shape) * 0.1
        if k.endswith("moving_mean"): p[k] = rng.uniform(0, 0.5, p[k].shape)
    x = rng.standard_normal((2, 9, 40))
    pt = tr.to_torch_params(p, False, torch.float64)
    for training in (True, False):
        out, stats = mo.xvector_2d_fwd(p, x, training=training)
        ref = tr.xvector_2d_fwd(pt, torch.tensor(x), training=training).numpy()
        assert out.shape == (2, 5) and np.abs(out - ref).max() < 1e-12
        assert np.allclose(np.exp(out).sum(-1), 1.0)
    a = np.maximum(rng.standard_normal((50, 7, 3)) * 2 + 1, 0)
    g, b = np.ones(3), np.zeros(3)
    y, mm, mv = mo.batchnorm_fwd(a, g, b, np.zeros(3), np.ones(3), True)
    assert np.abs(y.mean(axis=(0, 1))).max() < 1e-12
    assert np.abs(y.var(axis=(0, 1)) - a.var(axis=(0, 1)) / (a.var(axis=(0, 1)) + 1e-3)).max() < 1e-12
    assert np.allclose(mm, 0.01 * a.mean(axis=(0, 1))) and np.allclose(mv, 0.99 + 0.01 * a.var(axis=(0, 1), ddof=1))
    y2, mm2, mv2 = mo.batchnorm_fwd(a, g, b, mm, mv, False)
    assert mm2 is mm and np.allclose(y2, (a - mm) / np.sqrt(mv + 1e-3))


def test_optimizer_restatements_against_torch_and_closed_forms():
    """oracle/model_np.py sgd_step / rmsprop_step (the classes a config may name at keras_utils.py:137-140): SGD with momentum /
    Nesterov against torch.optim.SGD (same recurrences up to the sign convention of the velocity), RMSprop's momentum-free form
    against torch.optim.RMSprop (eps outside the root, as in that TensorFlow branch), and the fused form's epsilon-inside-the-root
    placement against its closed form after one step"""
    import torch
    from oracle import model_np as mo
    rng = np.random.default_rng(0)
    w = rng.standard_normal(50)
    grads = [rng.standard_normal(50) for _ in range(5)]
    for mom, nest in ((0.0, False), (0.9, False), (0.7, True)):
        p, vel = {"w": w.copy()}, {"w": np.zeros(50)}
        t = torch.tensor(w.copy(), requires_grad=True)
        opt = torch.optim.SGD([t], lr=0.05, momentum=mom, nesterov=nest)
        for g in grads:
            mo.sgd_step(p, {"w": g}, vel, lr=0.05, momentum=mom, nesterov=nest)
            t.grad = torch.tensor(g)
            opt.step()
        assert np.abs(p["w"] - t.detach().numpy()).max() < 1e-12, (mom, nest)
    for centered in (False, True):
        p, rms, mg, mm = {"w": w.copy()}, {"w": np.zeros(50)}, {"w": np.zeros(50)}, {"w": np.zeros(50)}
        t = torch.tensor(w.copy(), requires_grad=True)
        opt = torch.optim.RMSprop([t], lr=1e-2, alpha=0.9, eps=1e-7, centered=centered)
        for g in grads:
            mo.rmsprop_step(p, {"w": g}, rms, mg, mm, lr=1e-2, rho=0.9, eps=1e-7, centered=centered)
            t.grad = torch.tensor(g)
            opt.step()
        assert np.abs(p["w"] - t.detach().numpy()).max() < 1e-10, centered
    p, rms, mg, mm = {"w": w.copy()}, {"w": np.zeros(50)}, {"w": np.zeros(50)}, {"w": np.zeros(50)}
    mo.rmsprop_step(p, {"w": grads[0]}, rms, mg, mm, lr=1e-2, rho=0.9, momentum=0.5, eps=1e-3)
    assert np.abs(p["w"] - (w - 1e-2 * grads[0] / np.sqrt(0.1 * grads[0] ** 2 + 1e-3))).max() < 1e-15
    # from_logits=False cross-entropy: equals -log p_y inside the clip range, floor log(1 / (1 - 1e-7)) ~ 1e-7 when saturated
    z = rng.standard_normal((4, 5))
    y = np.array([0, 1, 2, 3])
    loss, dz = mo.sparse_ce_from_probs(z, y)
    lp = mo.log_softmax(z)
    assert abs(loss - float(-lp[np.arange(4), y].mean())) < 1e-12
    assert np.abs(dz - (mo.softmax(z) - np.eye(5)[y]) / 4).max() < 1e-12
    zs = np.array([[50.0, 0, 0]])
    ls, dzs = mo.sparse_ce_from_probs(zs, np.array([0]))
    assert 0 <= ls < 3e-7 and np.abs(dzs).max() == 0.0


def test_conv1d_padding_rules_follow_tensorflow():
    """model_np.conv1d_padding (Keras Conv1D padding as lidbox/models/cnn.py:25,33-36 passes it on): TensorFlow's documented
    example -- 13 inputs, filter 6, stride 5: VALID gives 2 outputs and drops the last two inputs, SAME gives 3 with one zero
    ahead and two behind -- plus the causal rule the x-vector uses, and the identities out_same = ceil(T / s),
    out_valid = ceil((T - k + 1) / s)"""
    from oracle import model_np as mo
    assert mo.conv1d_padding(13, 6, 5, 1, "valid") == (0, 0, 2)
    assert mo.conv1d_padding(13, 6, 5, 1, "same") == (1, 2, 3)
    assert mo.conv1d_padding(13, 6, 5, 1, "causal") == (5, 0, 3)
    assert mo.conv1d_padding(61, 7, 2, 1, "same") == (3, 3, 31)       # cnn.py conv_2 on an odd length
    assert mo.conv1d_padding(62, 7, 2, 1, "same") == (2, 3, 31)       # ... and an even one: one row less, the odd row behind
    assert mo.conv1d_padding(4, 7, 2, 1, "valid") == (0, 0, 0)
    for T in range(1, 40):
        for k in (1, 2, 5, 7):
            for s in (1, 2, 3):
                pl, pr, out = mo.conv1d_padding(T, k, s, 1, "same")
                assert out == -(-T // s) and pl + T + pr >= (out - 1) * s + k and pr - pl in (0, 1)
                assert mo.conv1d_padding(T, k, s, 1, "valid")[2] == max(0, -(-(T - k + 1) // s))
    x = np.arange(13, dtype=np.float64).reshape(1, 13, 1) + 1
    col = mo.im2col_causal(x, 6, 5, padding="same")[0]
    assert col[0].tolist() == [0, 1, 2, 3, 4, 5] and col[2].tolist() == [10, 11, 12, 13, 0, 0]
    assert mo.im2col_causal(x, 6, 5, padding="valid")[0, 1].tolist() == [6, 7, 8, 9, 10, 11]
