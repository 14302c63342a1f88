"""
BASELINE config 5 "bf16 compute / fp32 master": x-vector through the bf16-MFMA GEMM family.

Two kinds of check:
  * exact contract -- the forward pass must equal a float64 restatement in which every GEMM's two operands
    are rounded to bfloat16 first (and nothing else is): tolerance = fp32 summation round-off;
  * accuracy vs the unrounded float64 oracle at the tolerances SURVEY.md 8c states for the bf16 path:
    embedding cosine >= 0.999; loss and gradients within bf16's 2^-8 relative step accumulated over the
    five-layer chain (loss rel 2e-2, per-tensor gradient relative Frobenius error 5e-2).
"""
import os

import numpy as np
import pytest
import torch

from oracle import features_np as fo
from oracle import model_np as mo

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _dev(x, dtype=np.float32):
    return torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype))).cuda()


def _bf16(x):
    return torch.from_numpy(np.asarray(x, np.float32)).bfloat16().double().numpy()


def _cos(a, b):
    return (a * b).sum(-1) / (np.linalg.norm(a, axis=-1) * np.linalg.norm(b, axis=-1))


def _oracle_params(model):
    return {k: v.astype(np.float64) for k, v in model.get_weights().items()}


def _emulated_forward(p, x, dense16=False):
    """oracle forward with the bf16 contract: both operands of every Conv1D GEMM rounded, everything else float64 (the
    dense head computes in fp32; dense16: LIDBOX_BF16_DENSE=1 rounds its operands as well)"""
    r = _bf16 if dense16 else (lambda v: v)
    h = x
    for name, f, k, s in mo.XVECTOR_FRAMES:
        h = mo.conv1d_causal_fwd(_bf16(h), _bf16(p[name + ".W"]), p[name + ".b"], s)
    pooled = mo.stats_pool_fwd(h)
    emb = mo.dense_fwd(r(pooled), r(p["segment1.W"]), p["segment1.b"], relu=False)
    h1 = np.maximum(emb, 0)
    h2 = mo.dense_fwd(r(h1), r(p["segment2.W"]), p["segment2.b"], relu=True)
    z = mo.dense_fwd(r(h2), r(p["outputs.W"]), p["outputs.b"], relu=False)
    return mo.log_softmax(z), emb


@pytest.mark.parametrize("dense16", [False, True])
def test_bf16_xvector_forward_is_exactly_the_rounded_operand_model(dense16, monkeypatch):
    from lidbox_amd.models import xvector
    g = np.load(os.path.join(GOLDEN, "xvector_synth.npz"))
    monkeypatch.setenv("LIDBOX_BF16_DENSE", "1" if dense16 else "0")
    m = xvector.create((198, 40), 4, seed=0, compute_dtype="bfloat16")
    assert m.compute_dtype == "bfloat16" and m.flat.dtype == torch.float32        # fp32 master weights
    assert m.dense_gemm.name == ("bfloat16" if dense16 else "float32")
    x = g["logmel"]
    logp = m(_dev(x)).cpu().numpy()
    emb = xvector.as_embedding_extractor(m)(_dev(x)).cpu().numpy()
    ref_logp, ref_emb = _emulated_forward(_oracle_params(m), x.astype(np.float64), dense16)
    # fp32 hidden activations sit within an ulp of a bf16 rounding boundary now and then, so a handful of
    # elements round the other way than in float64; that moves outputs by ~1e-4 at most, far below the
    # 1e-2 .. 1e-1 distance between the bf16 and fp32 models measured next
    assert np.abs(emb - ref_emb).max() <= 2e-3 * np.abs(ref_emb).max()
    assert np.abs(logp - ref_logp).max() <= 2e-3
    # and the bf16 model is measurably NOT the fp32 model (the switch does something) yet stays within tolerance
    assert np.abs(logp - g["logp"]).max() > 1e-5
    assert _cos(emb, g["embedding"]).min() >= 0.999                                  # SURVEY.md 8c, bf16 path
    assert np.abs(logp - g["logp"]).max() <= 5e-2


def test_bf16_loss_and_gradients_close_to_float64_oracle():
    from lidbox_amd.models import xvector
    from lidbox_amd.train import Trainer
    g = np.load(os.path.join(GOLDEN, "xvector_synth.npz"))
    m = xvector.create((198, 40), 4, seed=0, compute_dtype="bf16")
    rng = np.random.default_rng(3)
    m.set_weights({k: rng.standard_normal(v.shape) * 0.05 for k, v in m.get_weights().items() if k.endswith(".b")})
    x, y = g["logmel"], g["labels"]
    loss, _ = Trainer(m, use_graph=False).loss_and_grads(_dev(x), _dev(y, np.int32))
    ref_loss, ref_g, _ = mo.xvector_loss_and_grads(_oracle_params(m), x.astype(np.float64), y)
    assert abs(float(loss) - ref_loss) <= 2e-2 * abs(ref_loss), (float(loss), ref_loss)
    for name in ref_g:
        got = m.param(name, grad=True).cpu().numpy().astype(np.float64)
        rel = np.linalg.norm(got - ref_g[name]) / np.linalg.norm(ref_g[name])
        assert rel <= 5e-2, (name, rel)


def test_bf16_requires_multiple_of_four_widths():
    from lidbox_amd.models import xvector
    with pytest.raises(ValueError):
        xvector.create((50, 40), 3, compute_dtype="bfloat16")
    with pytest.raises(ValueError):
        xvector.create((50, 40), 4, compute_dtype="float16")


@pytest.mark.parametrize("use_graph", [False, True])
def test_config5_bf16_train_step_tracks_fp32(use_graph):
    """BASELINE configs[4]: 100 languages, x-vector trunk -> segment1 -> L2 norm -> AP loss + C_avg, bf16 compute.
    The bf16 run must start at the oracle loss (rel 2e-2), decrease, and stay close to the fp32 run."""
    from lidbox_amd import _native as nv
    from lidbox_amd.features import audio
    from lidbox_amd.losses import SparseAngularProximity
    from lidbox_amd.metrics import SparseAverageDetectionCost
    from lidbox_amd.models import xvector
    from lidbox_amd.models.tdnn import DenseSpec, SequentialTDNN
    from lidbox_amd.testutil import synthetic_batch
    from lidbox_amd.train import Trainer
    N, D, B = 100, 512, 16
    sig, y = synthetic_batch(B, num_labels=N, duration_s=0.5)
    sd, yd = _dev(sig), _dev(y, np.int32)
    convs = [xvector.frame_layer(512, 5, 1, name="frame1"), xvector.frame_layer(512, 3, 2, name="frame2"),
             xvector.frame_layer(512, 3, 3, name="frame3"), xvector.frame_layer(512, 1, 1, name="frame4"),
             xvector.frame_layer(1500, 1, 1, name="frame5")]
    losses = {}
    for dt in ("float32", "bfloat16"):
        m = SequentialTDNN((48, 40), convs, "stats", [DenseSpec("segment1", D, relu=False)], output_activation=None, seed=0,
                           compute_dtype=dt)
        metric = SparseAverageDetectionCost(N, np.linspace(-np.pi, 0, 100))
        t = Trainer(m, loss=SparseAngularProximity(N, D), feature=dict(plan=audio.get_plan(16000, 400, 160), kind=nv.FEAT_LOGMEL),
                    use_graph=use_graph, metric=metric)
        losses[dt] = [float(t.train_step(sd, yd)) for _ in range(6)]
        assert 0.0 <= float(metric.result()) <= 1.0
        assert float(metric.tp.sum() + metric.fn.sum()) == 6 * B * 100
    feats = fo.extract_features(sig, [16000] * B, "logmelspectrogram")
    # step-0 oracle loss from a fresh initialisation (the trainers above have moved their weights)
    p0 = {k: v.astype(np.float64) for k, v in SequentialTDNN((48, 40), convs, "stats", [DenseSpec("segment1", D, relu=False)],
                                                                output_activation=None, seed=0).get_weights().items()}
    h = feats
    for name, f, k, s_ in mo.XVECTOR_FRAMES:
        h = mo.conv1d_causal_fwd(h, p0[name + ".W"], p0[name + ".b"], s_)
    z = mo.dense_fwd(mo.stats_pool_fwd(h), p0["segment1.W"], p0["segment1.b"], relu=False)
    ref = mo.ap_loss(y, mo.l2_normalize(z), N)
    lb, lf = losses["bfloat16"], losses["float32"]
    assert abs(lb[0] - ref) <= 2e-2 * abs(ref), (lb[0], ref)
    assert lb[-1] < lb[0]
    assert all(abs(a - b) <= 5e-2 * abs(b) for a, b in zip(lb, lf)), (lb, lf)


def test_bf16_storage_path_equals_fp32_source_path(monkeypatch):
    """the bf16-storage GEMMs (bf16 shadows of activations / gradients / weights: lidbox_gemm_bf16s_nt / _tn) compute what the
    fp32-source bf16 kernels compute: rounding happens where a shadow is written instead of where an operand is read.
    Four builds of the same model: shadows next to fp32 copies (LIDBOX_BF16_FP32_COPIES=1), shadows only with an fp32 copy of
    the last frame layer's output for the pooling (LIDBOX_BF16_POOL_FP32=1), shadows only (the default: no fp32 copy of any
    intermediate is written, the pooling reads the last layer's shadow), and the fp32-source kernels."""
    from lidbox_amd import _native as nv
    from lidbox_amd.features import audio
    from lidbox_amd.models import xvector
    from lidbox_amd.testutil import synthetic_batch
    from lidbox_amd.train import Trainer
    sig, y = synthetic_batch(16, num_labels=4, duration_s=1.0)
    sd = torch.from_numpy(sig).cuda()
    yd = torch.from_numpy(y.astype(np.int32)).cuda()
    plan = audio.get_plan(16000, 400, 160)
    res = {}
    for tag, storage, copies, pool32 in (("copies", "1", "1", "0"), ("only32", "1", "0", "1"), ("only", "1", "0", "0"), ("source", "0", "0", "0")):
        monkeypatch.setenv("LIDBOX_BF16_STORAGE", storage)
        monkeypatch.setenv("LIDBOX_BF16_FP32_COPIES", copies)
        monkeypatch.setenv("LIDBOX_BF16_POOL_FP32", pool32)
        m = xvector.create((98, 40), 4, seed=0, compute_dtype="bfloat16")
        assert m.bf16_storage == (storage == "1") and m.bf16_only == (tag in ("only", "only32"))
        t = Trainer(m, feature=dict(plan=plan, kind=nv.FEAT_LOGMEL), use_graph=(tag != "source"))
        loss, g = t.loss_and_grads(sd, yd)
        res[tag] = (float(loss), g.clone())
        ws = m.workspace(16, 98)
        if tag == "copies":
            # the shadows are what they claim to be
            for i in range(len(m.convs)):
                assert torch.equal(ws.act16[i], ws.act[i].bfloat16()), i
            for j in range(1, len(ws.dact16)):
                C = ws.dact[j].shape[2]
                assert j in ws.d16_fresh
                assert torch.equal(ws.dact16[j][:, :, :C], ws.dact[j].bfloat16()), j
                assert not ws.dact16[j][:, :, C:].any()                      # pad columns of a widened shadow stay zero
            # every conv input and every conv output gradient has a shadow; frame5's 1500 channels live in 1504-wide rows
            assert sum(a is not None for a in ws.act16) == 5 and sum(a is not None for a in ws.dact16) == 5
            assert ws.dact16[5].shape[2] == 1504
            # frame2 (k = 3 > s = 2) takes the output-stationary dgrad: its output gradient carries one zero trail row per
            # utterance, and the causal pad rows of dact[1] are never written
            assert ws.trail == [0, 0, 1, 0, 0, 0] and not ws.dact[2][:, -1, :].any() and not ws.dact[1][:, :2, :].any()
            assert ws.last16 is None
            shadows = [a.clone() for a in ws.act16[:5]] + [d.clone() for d in ws.dact16[1:]]
            last32 = ws.act[5].clone()
        elif tag == "only32":
            # same shadows bit for bit, and the fp32 copies of the intermediates were never touched
            now = list(ws.act16[:5]) + list(ws.dact16[1:])
            assert ws.last16 is None and all(torch.equal(a, b) for a, b in zip(shadows, now))
            assert all(not ws.act[j].any() for j in range(1, 5)) and all(not ws.dact[j].any() for j in range(1, 6))
        elif tag == "only":
            # the pooling reads frame5's shadow (1500 channels in 1504-wide rows) = bf16 of what the fp32 copy held; no fp32
            # activation behind the model input is written at all
            assert ws.last16 is not None and ws.last16.shape == (16, ws.Ts[-1], 1504)
            assert torch.equal(ws.last16[:, :, :1500], last32.bfloat16()) and not ws.last16[:, :, 1500:].any()
            assert all(not ws.act[j].any() for j in range(1, 5)) and all(not ws.dact[j].any() for j in range(1, 5))
            # the last layer's fp32 buffers are poisoned in this mode (nothing writes them; a reader that should have taken the shadow
            # sees NaN, not stale zeros -- ADVICE r5)
            assert bool(torch.isnan(ws.act[5]).all()) and bool(torch.isnan(ws.dact[5]).all())
            assert all(torch.equal(a, b) for a, b in zip(shadows[:5], ws.act16[:5]))
    # shadows-only changes where the ReLU masks are read (signs of the bf16 values): nothing else
    assert res["only32"][0] == res["copies"][0] and torch.equal(res["only32"][1], res["copies"][1])
    # the pooling over the shadow sees frame5's output rounded to bf16 -- as a Keras mixed_bfloat16 layer output is: one more
    # rounding of relative size 2^-9 per element ahead of a 33-frame mean
    assert abs(res["only"][0] - res["only32"][0]) <= 2e-4 * abs(res["only32"][0])
    assert float(torch.linalg.norm(res["only"][1] - res["only32"][1]) / torch.linalg.norm(res["only32"][1])) <= 1e-2
    # the storage and fp32-source paths accumulate K in different chunk orders (64- vs 32-deep tiles), so an activation can
    # land on the other side of a bf16 rounding boundary here and there: agreement far below one bf16 step (4e-3), not bit
    # equality; the bias gradients of the storage path are summed from the shadows
    assert abs(res["only"][0] - res["source"][0]) <= 1e-4 * abs(res["source"][0])
    ga, gb = res["only"][1], res["source"][1]
    assert float(torch.linalg.norm(ga - gb) / torch.linalg.norm(gb)) <= 1e-2          # vs 5e-2 allowed against the float64 oracle
    # and the captured train step learns on the storage path
    monkeypatch.setenv("LIDBOX_BF16_STORAGE", "1")
    t = Trainer(xvector.create((98, 40), 4, seed=0, compute_dtype="bfloat16"), feature=dict(plan=plan, kind=nv.FEAT_LOGMEL))
    l0 = float(t.train_step(sd, yd))
    for _ in range(12):
        l1 = float(t.train_step(sd, yd))
    assert np.isfinite(l1) and l1 < l0


def test_bf16_models_without_qualifying_layers_keep_the_fp32_source_kernels():
    """the CNN's 12- / 500-channel layers cannot be read as 16-byte bf16 pieces: no shadows are allocated and the bf16
    compute path still trains (fp32-source kernels)"""
    from lidbox_amd.models import cnn
    from lidbox_amd.train import Trainer
    rng = np.random.default_rng(31)
    m = cnn.create((40, 12), 4, seed=1, compute_dtype="bfloat16")
    assert not m.bf16_storage and not hasattr(m, "flat16")
    x = torch.from_numpy(rng.standard_normal((8, 40, 12)).astype(np.float32)).cuda()
    y = torch.from_numpy(rng.integers(0, 4, size=8).astype(np.int32)).cuda()
    t = Trainer(m, use_graph=True)
    l0 = float(t.train_step(x, y))
    for _ in range(10):
        l1 = float(t.train_step(x, y))
    assert np.isfinite(l1) and l1 < l0


@pytest.mark.parametrize("T", [61, 7, 2])
def test_bf16_storage_extended_xvector_strided_layers(T, monkeypatch):
    """the ten-layer x-vector on the shadows: frame3 (k 3, s 2, followed by a k = 1 layer: its output-stationary dgrad reads
    the zero trail row), frame5 (k = s = 3), frame7 (k 3 < s 4: rows no tap writes), odd / tiny frame counts.  Storage path
    (shadows only) against the fp32-source bf16 kernels, and repeated backward passes give the same bits (nothing
    accumulates across steps in rows that are never cleared)."""
    from lidbox_amd.models import xvector_extended
    from lidbox_amd.train import Trainer
    rng = np.random.default_rng(T)
    B = 6
    x = _dev(rng.standard_normal((B, T, 40)))
    y = _dev(rng.integers(0, 4, size=B), np.int32)
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("LIDBOX_BF16_STORAGE", flag)
        m = xvector_extended.create((T, 40), 4, seed=3, compute_dtype="bfloat16")
        assert m.bf16_storage == (flag == "1") and m.bf16_only == (flag == "1")
        t = Trainer(m, use_graph=False)
        loss, g = t.loss_and_grads(x, y)
        res[flag] = (float(loss), g.clone())
        if flag == "1":
            ws = m.workspace(B, T)
            assert ws.trail[3] == 1 and sum(ws.trail) == 1          # only frame3's output gradient needs a trail row
            loss2, g2 = t.loss_and_grads(x, y)
            assert float(loss2) == float(loss) and torch.equal(g2, res["1"][1])
    # ten rounded layers deep, the two paths' different K chunk orders flip more roundings than in the five-layer net (measured
    # 2e-2 in norm at T = 61); a wrong tap, a missed trail row or a stale hole row would show as O(1)
    assert abs(res["1"][0] - res["0"][0]) <= 1e-3 * abs(res["0"][0])
    ga, gb = res["1"][1], res["0"][1]
    assert float(torch.linalg.norm(ga - gb) / torch.linalg.norm(gb)) <= 5e-2
    # and both stay within the bf16 tolerance of the fp32 model
    monkeypatch.setenv("LIDBOX_BF16_STORAGE", "1")
    m32 = xvector_extended.create((T, 40), 4, seed=3)
    l32, g32 = Trainer(m32, use_graph=False).loss_and_grads(x, y)
    assert abs(res["1"][0] - float(l32)) <= 2e-2 * abs(float(l32))
    ea = float(torch.linalg.norm(ga - g32) / torch.linalg.norm(g32))
    eb = float(torch.linalg.norm(gb - g32) / torch.linalg.norm(g32))
    # ten bf16-rounded layers at random initialisation: ~9 % in norm for either path (measured 9.2 / 9.0 % at T = 61); the
    # storage path must not be further from fp32 than the fp32-source path is, beyond noise
    assert ea <= 0.15 and eb <= 0.15 and ea <= 1.25 * eb + 1e-3, (ea, eb)


def test_bf16_storage_with_a_first_layer_that_does_not_qualify(monkeypatch):
    """12 MFCC channels: conv 0 cannot read 16-byte bf16 pieces, so it stays on the fp32-source kernels while the other four
    layers run on the shadows (its output and its wgrad operands are converted on the way); the fp32 copies are kept"""
    from lidbox_amd.models import xvector
    from lidbox_amd.train import Trainer
    rng = np.random.default_rng(12)
    B, T = 8, 50
    x = _dev(rng.standard_normal((B, T, 12)))
    y = _dev(rng.integers(0, 4, size=B), np.int32)
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("LIDBOX_BF16_STORAGE", flag)
        m = xvector.create((T, 12), 4, seed=5, compute_dtype="bfloat16")
        assert m.bf16_storage == (flag == "1") and not m.bf16_only
        t = Trainer(m, use_graph=False)
        loss, g = t.loss_and_grads(x, y)
        res[flag] = (float(loss), g.clone())
        if flag == "1":
            ws = m.workspace(B, T)
            assert ws.act16[0] is None and ws.act16[1] is not None and ws.dact16[1] is None and ws.dact16[2] is not None
            assert ws.act[2].any() and ws.dact[2].any()              # fp32 copies written
    assert abs(res["1"][0] - res["0"][0]) <= 1e-4 * abs(res["0"][0])
    assert float(torch.linalg.norm(res["1"][1] - res["0"][1]) / torch.linalg.norm(res["0"][1])) <= 1e-2
    monkeypatch.setenv("LIDBOX_BF16_STORAGE", "1")
    t = Trainer(xvector.create((T, 12), 4, seed=5, compute_dtype="bfloat16"))
    l0 = float(t.train_step(x, y))
    for _ in range(10):
        l1 = float(t.train_step(x, y))
    assert np.isfinite(l1) and l1 < l0
