"""
GPU parity of the x-vector / CNN models, embedding extractor and the train step against the
oracle and the committed golden vectors.

Tolerances: embedding cosine >= 0.9999, loss rel <= 1e-4, log-prob max-abs 1e-3,
per-layer gradient rel 1e-3 (fp32 MFMA chains over K up to 50k vs float64).
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


def _cos(a, b):
    return (a * b).sum(-1) / (np.linalg.norm(a, axis=-1) * np.linalg.norm(b, axis=-1))


def _oracle_params(model):
    return {k: v.astype(np.float64) for k, v in model.get_weights().items()}


def test_xvector_param_count_and_layout():
    from lidbox_amd.models import xvector
    m = xvector.create((198, 40), 4, seed=0)
    assert m.count_params() == 4510176
    assert xvector.create((198, 40), 100, seed=0).count_params() == 4559424
    w = m.get_weights()
    assert w["frame1.W"].shape == (5, 40, 512) and w["frame5.W"].shape == (1, 512, 1500)
    assert w["segment1.W"].shape == (3000, 512) and w["outputs.W"].shape == (512, 4)
    # same seed as the oracle initialiser -> identical weights (glorot_uniform, zero bias)
    p = mo.xvector_init(40, 4, seed=0)
    for k in p:
        assert np.array_equal(w[k], p[k]), k


def test_xvector_forward_matches_golden_and_oracle():
    from lidbox_amd.models import xvector
    g = np.load(os.path.join(GOLDEN, "xvector_synth.npz"))
    m = xvector.create((198, 40), 4, seed=0)
    x = _dev(g["logmel"])
    logp = m(x, training=False).cpu().numpy()
    assert logp.shape == (4, 4)
    assert np.abs(logp - g["logp"]).max() < 1e-3
    emb = xvector.as_embedding_extractor(m)(x).cpu().numpy()
    assert emb.shape == (4, 512)
    assert _cos(emb, g["embedding"]).min() >= 0.9999
    assert np.abs(m(x, training=True).cpu().numpy() - logp).max() == 0     # no dropout -> same path


def test_waveform_to_embedding_end_to_end():
    """waveform -> fused log-mel kernel -> x-vector, vs the float64 oracle chain"""
    from lidbox_amd.data import tf_utils
    from lidbox_amd.models import xvector
    from lidbox_amd.testutil import synthetic_batch
    sig, y = synthetic_batch(4, num_labels=4)
    g = np.load(os.path.join(GOLDEN, "xvector_synth.npz"))
    assert np.array_equal(y, g["labels"])
    feats = tf_utils.extract_features(_dev(sig), [16000] * 4, "logmelspectrogram")
    assert np.abs(feats.cpu().numpy() - g["logmel"]).max() <= 1e-3
    m = xvector.create((198, 40), 4, seed=0)
    emb = xvector.as_embedding_extractor(m)(feats).cpu().numpy()
    assert _cos(emb, g["embedding"]).min() >= 0.9999


def test_xvector_loss_and_gradients_match_oracle():
    from lidbox_amd.models import xvector
    from lidbox_amd.train import Trainer
    g = np.load(os.path.join(GOLDEN, "xvector_synth.npz"))
    m = xvector.create((198, 40), 4, seed=0)
    # non-zero biases so every bias path carries signal
    rng = np.random.default_rng(3)
    m.set_weights({k: rng.standard_normal(v.shape) * 0.05 for k, v in m.get_weights().items() if k.endswith(".b")})
    tr = Trainer(m, use_graph=False)
    x, y = g["logmel"], g["labels"]
    loss, grads = tr.loss_and_grads(_dev(x), _dev(y, np.int32))
    ref_loss, ref_g, _ = mo.xvector_loss_and_grads(_oracle_params(m), x.astype(np.float64), y)
    assert abs(float(loss) - ref_loss) <= 1e-4 * abs(ref_loss)
    for name in ref_g:
        got = m.param(name, grad=True).cpu().numpy()
        scale = np.abs(ref_g[name]).max()
        assert np.abs(got - ref_g[name]).max() <= 1e-3 * scale, name
    # golden first-step loss and gradient norms (regression pin; zero-bias weights as in make_golden.py)
    m0 = xvector.create((198, 40), 4, seed=0)
    loss0, _ = Trainer(m0, use_graph=False).loss_and_grads(_dev(x), _dev(y, np.int32))
    assert abs(float(loss0) - float(g["loss"])) <= 1e-4 * float(g["loss"])
    for name, norm in zip(g["grad_names"], g["grad_norms"]):
        got = float(m0.param(str(name), grad=True).norm())
        assert abs(got - norm) <= 1e-3 * max(norm, 1e-6), name


@pytest.mark.parametrize("kw", [dict(overlap_wgrad=True), dict(overlap_head_wgrad=True)])
@pytest.mark.parametrize("use_graph", [False, True])
def test_side_stream_wgrad_options_give_the_same_gradients(kw, use_graph):
    """wgrad GEMMs on a second HIP stream (all of them, or the dense head's only): same bits as the single-stream step,
    eager and inside a captured graph (fork / join edges)"""
    from lidbox_amd.models import xvector
    from lidbox_amd.train import Trainer
    g = np.load(os.path.join(GOLDEN, "xvector_synth.npz"))
    x, y = _dev(g["logmel"]), _dev(g["labels"], np.int32)
    ref_m = xvector.create((198, 40), 4, seed=0)
    ref_t = Trainer(ref_m, use_graph=False)
    ref_losses = [float(ref_t.train_step(x, y)) for _ in range(3)]
    m = xvector.create((198, 40), 4, seed=0)
    t = Trainer(m, use_graph=use_graph, **kw)
    losses = [float(t.train_step(x, y)) for _ in range(3)]
    assert losses == ref_losses and torch.equal(m.flat, ref_m.flat)


def test_fused_output_layer_equals_the_separate_launches(monkeypatch):
    """Trainer fuses the last Dense + log_softmax + cross-entropy and their backward (lidbox_softmax_head_fwd_bwd) when the
    model allows it; LIDBOX_NO_FUSED_OUTPUT=1 keeps the GEMM / log-softmax / NLL launches: same loss and gradients to fp32
    summation order, same trajectory over a few Adam steps"""
    from lidbox_amd.models import cnn, xvector
    from lidbox_amd.train import Trainer
    g = np.load(os.path.join(GOLDEN, "xvector_synth.npz"))
    x, y = _dev(g["logmel"]), _dev(g["labels"], np.int32)
    res = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("LIDBOX_NO_FUSED_OUTPUT", flag)
        m = xvector.create((198, 40), 4, seed=0)
        t = Trainer(m, use_graph=True)
        assert t.fuse_output == (flag == "0")
        loss, grads = t.loss_and_grads(x, y)
        res[flag] = (float(loss), grads.clone(), [float(t.train_step(x, y)) for _ in range(3)])
    assert abs(res["0"][0] - res["1"][0]) <= 1e-6 * abs(res["1"][0])
    assert float(torch.linalg.norm(res["0"][1] - res["1"][1]) / torch.linalg.norm(res["1"][1])) <= 1e-5
    # Adam divides by sqrt(v) ~ |g|: once the loss collapses (third step of this separable toy batch) the last bits of the
    # gradients are amplified, so only the first steps are compared tightly
    assert np.allclose(res["0"][2][:2], res["1"][2][:2], rtol=1e-5) and np.allclose(res["0"][2][2], res["1"][2][2], rtol=2e-2)
    # models the fusion does not apply to keep the separate launches: 100 classes, a single Dense
    monkeypatch.setenv("LIDBOX_NO_FUSED_OUTPUT", "0")
    assert not Trainer(xvector.create((50, 40), 100, seed=0)).fuse_output
    assert Trainer(cnn.create((40, 12), 4, seed=1)).fuse_output == (len(cnn.create((40, 12), 4, seed=1).denses) >= 2)


def test_train_steps_match_oracle_adam_and_graph_equals_eager():
    from lidbox_amd.models import xvector
    from lidbox_amd.train import Trainer
    rng = np.random.default_rng(4)
    x = rng.standard_normal((6, 50, 24)).astype(np.float32)
    y = rng.integers(0, 5, size=6).astype(np.int32)
    xd, yd = _dev(x), _dev(y, np.int32)
    m_e = xvector.create((50, 24), 5, seed=7)
    m_g = xvector.create((50, 24), 5, seed=7)
    p = _oracle_params(m_e)
    am = {k: np.zeros_like(v) for k, v in p.items()}
    av = {k: np.zeros_like(v) for k, v in p.items()}
    t_e, t_g = Trainer(m_e, use_graph=False), Trainer(m_g, use_graph=True)
    for step in range(1, 4):
        le = float(t_e.train_step(xd, yd))
        lg = float(t_g.train_step(xd, yd))
        lo, go, _ = mo.xvector_loss_and_grads(p, x.astype(np.float64), y)
        mo.adam_step(p, go, am, av, step)
        assert abs(le - lo) <= 2e-4 * abs(lo), (step, le, lo)
        assert le == lg, (step, le, lg)                       # graph replay == eager, bit for bit
    assert torch.equal(m_e.flat, m_g.flat)
    assert t_e.step_count == 3 and t_g.step_count == 3
    for k, v in m_e.get_weights().items():
        # Adam normalises the update, so after 3 steps weights moved ~3e-3; compare the movement
        assert np.abs(v - p[k]).max() <= 3e-4, k
    # the loss goes down on a fixed batch
    l0 = float(t_e.train_step(xd, yd))
    for _ in range(10):
        l1 = float(t_e.train_step(xd, yd))
    assert l1 < l0


@pytest.mark.parametrize("opt", [dict(cls="SGD", lr=0.05), dict(cls="SGD", lr=0.02, momentum=0.9), dict(cls="SGD", lr=0.02, momentum=0.8, nesterov=True),
                                 dict(cls="RMSprop", lr=2e-3), dict(cls="RMSprop", lr=1e-3, momentum=0.5, rho=0.8),
                                 dict(cls="RMSprop", lr=1e-3, centered=True, epsilon=1e-6), dict(cls="RMSprop", lr=1e-4, centered=True, momentum=0.5)])
@pytest.mark.parametrize("use_graph", [False, True])
def test_train_steps_match_oracle_sgd_and_rmsprop(opt, use_graph):
    """the optimizer classes a config may name besides Adam (reference keras_utils.py:137-140: getattr(tf.keras.optimizers, cls)):
    four steps of lidbox_sgd_step / lidbox_rmsprop_step inside the train step against the oracle's restatement of the TensorFlow
    2.3 dense updates (momentum / Nesterov, RMSprop's two epsilon placements, centered)"""
    from lidbox_amd.models import xvector
    from lidbox_amd.train import Trainer
    rng = np.random.default_rng(5)
    x = rng.standard_normal((6, 50, 24)).astype(np.float32)
    y = rng.integers(0, 5, size=6).astype(np.int32)
    xd, yd = _dev(x), _dev(y, np.int32)
    m = xvector.create((50, 24), 5, seed=7)
    p = _oracle_params(m)
    w0 = {k: v.copy() for k, v in p.items()}
    bufs = [{k: np.zeros_like(v) for k, v in p.items()} for _ in range(3)]
    tr = Trainer(m, optimizer=opt, use_graph=use_graph)
    kw = {k: v for k, v in opt.items() if k != "cls"}
    for step in range(1, 5):
        l = float(tr.train_step(xd, yd))
        lo, go, _ = mo.xvector_loss_and_grads(p, x.astype(np.float64), y)
        if opt["cls"] == "SGD":
            mo.sgd_step(p, go, bufs[0], **kw)
        else:
            mo.rmsprop_step(p, go, bufs[0], bufs[1], bufs[2], **{("eps" if k == "epsilon" else k): v for k, v in kw.items()})
        assert abs(l - lo) <= (5e-4 if opt["cls"] == "SGD" else 3e-3) * abs(lo), (step, l, lo)
    assert tr.step_count == 4
    for k, v in m.get_weights().items():
        moved = np.abs(p[k] - w0[k]).max()
        # RMSprop normalises every update to ~lr / sqrt(1 - rho) (like Adam): a weight whose gradient is ~0 may move the other
        # way on an fp32 summation-order difference, so the bulk is held tight and the worst element to a fraction of the movement
        # (the update arithmetic itself is pinned to 1e-6 in tests/test_ops_gpu.py::test_sgd_and_rmsprop_updates)
        err = np.abs(v - p[k])
        if opt["cls"] == "SGD":
            assert err.max() <= 2e-2 * moved + 1e-7, (k, err.max(), moved)
        else:
            assert np.median(err) <= 1e-3 * moved + 1e-7 and err.max() <= 0.25 * moved + 1e-7, (k, np.median(err), err.max(), moved)
    with pytest.raises(ValueError):
        Trainer(m, optimizer=dict(cls="Adagrad"))
    with pytest.raises(ValueError):
        Trainer(m, optimizer=dict(cls="SGD", beta_1=0.9))


def test_softmax_output_and_crossentropy_on_probabilities():
    """cnn.create(output_activation="softmax") (reference cnn.py:43-44: getattr(tf.nn, output_activation)) with Keras
    SparseCategoricalCrossentropy(from_logits=False): probabilities out of the model, the clipped-probability loss and its
    gradients against the oracle (oracle/model_np.py: sparse_ce_from_probs; torch autograd through the same formula)"""
    from lidbox_amd.models import cnn
    from lidbox_amd.train import Trainer
    from oracle import torch_ref as tr
    rng = np.random.default_rng(10)
    x = rng.standard_normal((5, 61, 12))
    y = rng.integers(0, 4, size=5).astype(np.int32)
    m = cnn.create((61, 12), 4, output_activation="softmax", seed=2)
    p = _oracle_params(m)
    probs = m(_dev(x)).cpu().numpy()
    assert np.abs(probs - np.exp(mo.cnn_fwd(p, x))).max() < 1e-5 and np.abs(probs.sum(axis=1) - 1).max() < 1e-6
    pt = tr.to_torch_params(p, True, torch.float64)
    logp = tr.cnn_fwd(pt, torch.tensor(x))
    q = torch.clamp(torch.exp(logp), 1e-7, 1 - 1e-7)
    yt = torch.tensor(y.astype(np.int64))
    ref = (torch.log(q.sum(dim=1)) - torch.log(q.gather(1, yt[:, None])[:, 0])).mean()
    ref.backward()
    t = Trainer(m, loss="sparse_categorical_crossentropy_probs", use_graph=False)
    assert not t.fuse_output
    loss, _ = t.loss_and_grads(_dev(x), _dev(y, np.int32))
    assert abs(float(loss) - float(ref.detach())) <= 1e-5 * abs(float(ref.detach()))
    for k in p:
        ref_g = pt[k].grad.numpy()
        assert np.abs(m.param(k, grad=True).cpu().numpy() - ref_g).max() <= 1e-3 * max(1e-12, np.abs(ref_g).max()), k
    # the loss op alone against the numpy restatement, including saturated rows (a clipped probability passes no gradient)
    from lidbox_amd import _native as nv
    z = rng.standard_normal((7, 6)).astype(np.float32) * 3
    z[0] = [40, 0, 0, 0, 0, 0]                                   # p_0 = 1 - O(1e-17): clipped, zero gradient row
    z[1] = [-30, 0, 0, 0, 0, 1]                                  # p_0 < 1e-7: its own term is clipped
    yy = np.array([0, 0, 3, 1, 5, 2, 4], np.int32)
    zd, dz, pr, lo = _dev(z), torch.zeros(7, 6, device="cuda"), torch.zeros(7, 6, device="cuda"), torch.zeros(1, device="cuda")
    nv.check(nv.lib.lidbox_softmax_nll_fwd_bwd(nv.ptr(zd), nv.ptr(_dev(yy, np.int32)), 7, 6, 1.0 / 7, nv.ptr(pr), nv.ptr(lo), nv.ptr(dz), nv.current_stream()))
    rl, rdz = mo.sparse_ce_from_probs(z, yy)
    assert abs(float(lo) - rl) <= 1e-5 * abs(rl) and np.abs(dz.cpu().numpy() - rdz).max() <= 1e-6
    assert np.abs(dz.cpu().numpy()[0]).max() <= 1e-9 and np.abs(pr.cpu().numpy() - mo.softmax(z.astype(np.float64))).max() < 1e-6
    # a loss / model mismatch is refused; from_logits=True still needs the log-softmax outputs
    with pytest.raises(ValueError):
        Trainer(cnn.create((61, 12), 4, seed=2), loss="sparse_categorical_crossentropy_probs")
    with pytest.raises(ValueError):
        Trainer(m, loss="sparse_categorical_crossentropy")


def test_train_step_from_waveforms_writes_features_in_place():
    from lidbox_amd import _native as nv
    from lidbox_amd.features import audio
    from lidbox_amd.models import xvector
    from lidbox_amd.testutil import synthetic_batch
    from lidbox_amd.train import Trainer
    sig, y = synthetic_batch(8, num_labels=4)
    m1, m2 = xvector.create((198, 40), 4, seed=0), xvector.create((198, 40), 4, seed=0)
    plan = audio.get_plan(16000, 400, 160)
    t1 = Trainer(m1, feature=dict(plan=plan, kind=nv.FEAT_LOGMEL), use_graph=True)
    t2 = Trainer(m2, use_graph=False)
    sd, yd = _dev(sig), _dev(y, np.int32)
    feats = plan.run(nv.FEAT_LOGMEL, sd)
    for _ in range(2):
        a = float(t1.train_step(sd, yd))
        b = float(t2.train_step(feats, yd))
        assert a == b
    assert torch.equal(m1.flat, m2.flat)


@pytest.mark.parametrize("use_graph,kind,dtype", [(True, "logmel", "float32"), (False, "logmel", "float32"), (True, "mfcc_cmvn", "float32"),
                                                  (True, "logmel", "bfloat16")])
def test_feature_prefetch_equals_in_line_extraction(use_graph, kind, dtype):
    """Trainer.train_step(next_inputs=): the next batch's features are extracted during the running step on a second stream
    into the other half of the double-buffered model input (the tf.data prefetch of lidbox/data/steps.py:725-736).  Rotating
    over three batches, losses and weights equal the in-line path bit for bit; a batch that was NOT announced is extracted in
    line (no stale buffer is ever consumed)."""
    from lidbox_amd import _native as nv
    from lidbox_amd.features import audio
    from lidbox_amd.models import cnn, xvector
    from lidbox_amd.testutil import synthetic_batch
    from lidbox_amd.train import Trainer
    plan = audio.get_plan(16000, 400, 160)
    if kind == "logmel":
        mk = lambda: xvector.create((98, 40), 4, seed=0, compute_dtype=dtype)
        feature = dict(plan=plan, kind=nv.FEAT_LOGMEL)
    else:
        mk = lambda: cnn.create((98, 12), 4, seed=0)
        feature = dict(plan=plan, kind=nv.FEAT_MFCC, cmvn=True)
    batches = []
    for i in range(3):
        sig, y = synthetic_batch(6, num_labels=4, duration_s=1.0, seed=40 + i)
        batches.append((_dev(sig), _dev(y, np.int32)))
    m1, m2 = mk(), mk()
    t1 = Trainer(m1, feature=dict(feature), use_graph=use_graph)
    t2 = Trainer(m2, feature=dict(feature), use_graph=use_graph)
    order = [0, 1, 2, 0, 1, 2, 2, 0, 1]                   # the repeated 2 was announced as 0: must fall back to in-line extraction
    announced = [1, 2, 0, 1, 2, 0, 0, 1, 2]
    for i, a in zip(order, announced):
        la = float(t1.train_step(*batches[i], next_inputs=batches[a][0]))
        lb = float(t2.train_step(*batches[i]))
        assert la == lb, (i, la, lb)
    assert torch.equal(m1.flat, m2.flat)
    # a probe in between invalidates the prefetched buffer instead of consuming it
    t1.loss_and_grads(*batches[1])
    la, lb = float(t1.train_step(*batches[2], next_inputs=batches[0][0])), float(t2.train_step(*batches[2]))
    assert la == lb and torch.equal(m1.flat, m2.flat)
    with pytest.raises(ValueError):
        Trainer(mk(), use_graph=False).train_step(torch.zeros((6, 98, 40 if kind == "logmel" else 12), device="cuda"), batches[0][1],
                                                  next_inputs=batches[0][0])


@pytest.mark.parametrize("shape", [(1, 1, 1), (3, 1, 7), (2, 5, 40), (4, 50, 13), (10, 400, 100), (2, 198, 40)])
def test_models_valid_output_any_shape(shape):
    """reference tests/test_models.py:30-35,65-68,104-107: [B,num_outputs], no NaN, training in {F,T}"""
    from lidbox_amd.models import cnn, xvector
    rng = np.random.default_rng(8)
    x = _dev(rng.uniform(-1e3, 1e3, size=shape))
    for module in (xvector, cnn):
        for n_out in (1, 100):
            m = module.create(shape[1:], n_out, seed=1)
            for t in (False, True):
                yv = m(x, training=t).cpu().numpy()
                assert yv.shape == (shape[0], n_out) and not np.isnan(yv).any()
    m = xvector.create(shape[1:], 3, channel_dropout_rate=0.5, seed=1)
    assert not torch.isnan(m(x, training=True)).any()


def test_cnn_matches_oracle_forward_and_backward():
    from lidbox_amd.models import cnn
    from lidbox_amd.train import Trainer
    rng = np.random.default_rng(9)
    x = rng.standard_normal((3, 61, 12))
    y = rng.integers(0, 4, size=3).astype(np.int32)
    m = cnn.create((61, 12), 4, seed=2)
    p = _oracle_params(m)
    ref = mo.cnn_fwd(p, x)
    got = m(_dev(x)).cpu().numpy()
    assert np.abs(got - ref).max() < 1e-3
    emb = cnn.as_embedding_extractor(m)(_dev(x)).cpu().numpy()
    assert _cos(emb, mo.cnn_fwd(p, x, embedding=True)).min() >= 0.9999
    # gradients vs torch autograd on the CPU restatement
    from oracle import torch_ref as tr
    pt = tr.to_torch_params(p, True, torch.float64)
    tr.sparse_ce_from_logits(tr.cnn_fwd(pt, torch.tensor(x)), torch.tensor(y.astype(np.int64))).backward()
    loss, _ = Trainer(m, use_graph=False).loss_and_grads(_dev(x), _dev(y, np.int32))
    for k in p:
        ref_g = pt[k].grad.numpy()
        got_g = m.param(k, grad=True).cpu().numpy()
        assert np.abs(got_g - ref_g).max() <= 1e-3 * max(1e-12, np.abs(ref_g).max()), k


@pytest.mark.parametrize("padding", ["valid", "same", "causal"])
@pytest.mark.parametrize("T", [61, 62, 12])
@pytest.mark.parametrize("compute_dtype", ["float32", "bfloat16"])
def test_cnn_padding_modes_match_oracle(padding, T, compute_dtype):
    """cnn.create(padding=...) (reference cnn.py:25,33-36 hands it to every Conv1D): TensorFlow's "valid" / "same" geometry --
    odd and even lengths against the stride-2 layer, an input barely longer than the kernels -- forward, embedding, every
    gradient (the grouped dgrad writes rows in padded coordinates; untouched rows must read as zero), and a captured step"""
    from lidbox_amd.models import cnn
    from lidbox_amd.train import Trainer
    from oracle import torch_ref as tr
    rng = np.random.default_rng(T)
    x = rng.standard_normal((3, T, 12))
    y = rng.integers(0, 4, size=3).astype(np.int32)
    m = cnn.create((T, 12), 4, padding=padding, seed=2, compute_dtype=compute_dtype)
    p = _oracle_params(m)
    tol = 1e-3 if compute_dtype == "float32" else 4e-2
    ref = mo.cnn_fwd(p, x, padding=padding)
    got = m(_dev(x)).cpu().numpy()
    assert got.shape == ref.shape and np.abs(got - ref).max() < tol
    emb = cnn.as_embedding_extractor(m)(_dev(x)).cpu().numpy()
    assert _cos(emb, mo.cnn_fwd(p, x, embedding=True, padding=padding)).min() >= (0.9999 if compute_dtype == "float32" else 0.999)
    pt = tr.to_torch_params(p, True, torch.float64)
    ref_loss = tr.sparse_ce_from_logits(tr.cnn_fwd(pt, torch.tensor(x), padding=padding), torch.tensor(y.astype(np.int64)))
    ref_loss.backward()
    t = Trainer(m, use_graph=False)
    loss, _ = t.loss_and_grads(_dev(x), _dev(y, np.int32))
    assert abs(float(loss) - float(ref_loss.detach())) < tol
    for k in p:
        ref_g = pt[k].grad.numpy()
        got_g = m.param(k, grad=True).cpu().numpy()
        if compute_dtype == "float32":
            assert np.abs(got_g - ref_g).max() <= 1e-3 * max(1e-12, np.abs(ref_g).max()), k
        else:       # bf16 operands, per-tensor Frobenius error: 3 utterances through 7 layers leave conv_1.W at 0.10 - 0.12 in every
            #         padding mode, "causal" (the reference's default) included -- a geometry error would be O(1)
            assert np.linalg.norm(got_g - ref_g) <= 0.2 * np.linalg.norm(ref_g), (k, np.linalg.norm(got_g - ref_g) / np.linalg.norm(ref_g))
    # a second pass gives the same gradients (rows a pass leaves untouched are not polluted by the one before)
    g1 = {k: m.param(k, grad=True).clone() for k in p}
    t.loss_and_grads(_dev(x), _dev(y, np.int32))
    assert all(torch.equal(g1[k], m.param(k, grad=True)) for k in p)
    tg = Trainer(cnn.create((T, 12), 4, padding=padding, seed=2, compute_dtype=compute_dtype))
    l0 = float(tg.train_step(_dev(x), _dev(y, np.int32)))
    l1 = float(tg.train_step(_dev(x), _dev(y, np.int32)))
    assert abs(l0 - float(ref_loss.detach())) < tol and l1 < l0
    with pytest.raises(ValueError):
        cnn.create((T, 12), 4, padding="full")
    # xvector.frame_layer hands `padding` to Conv1D too (reference xvector.py:38-39)
    from lidbox_amd.models import xvector
    assert xvector.frame_layer(8, 7, 2, padding=padding).geometry(T) == mo.conv1d_padding(T, 7, 2, 1, padding)


def test_angular_proximity_head_train_step():
    """config 5 head: trunk -> segment1 (affine) -> L2 normalise -> SparseAngularProximity + C_avg"""
    from lidbox_amd.losses import SparseAngularProximity
    from lidbox_amd.metrics import SparseAverageDetectionCost
    from lidbox_amd.models import xvector
    from lidbox_amd.models.tdnn import DenseSpec, SequentialTDNN
    from lidbox_amd.train import Trainer
    from oracle import torch_ref as tr
    rng = np.random.default_rng(10)
    B, T, C, N, D = 8, 40, 16, 5, 32
    x = rng.standard_normal((B, T, C))
    y = rng.integers(0, N, size=B).astype(np.int32)
    convs = [xvector.frame_layer(32, 5, 1, name="frame1"), xvector.frame_layer(32, 3, 2, name="frame2")]
    m = SequentialTDNN((T, C), convs, "stats", [DenseSpec("segment1", D, relu=False)], output_activation=None, seed=3)
    metric = SparseAverageDetectionCost(N, np.linspace(-np.pi, 0, 20))
    t = Trainer(m, loss=SparseAngularProximity(N, D), use_graph=False, metric=metric)
    loss, _ = t.loss_and_grads(_dev(x), _dev(y, np.int32))
    # torch autograd reference of the same head
    p = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in m.get_weights().items()}
    h = torch.tensor(x)
    h = tr.conv1d_causal(h, p["frame1.W"], p["frame1.b"], 1)
    h = tr.conv1d_causal(h, p["frame2.W"], p["frame2.b"], 2)
    z = tr.stats_pool(h) @ p["segment1.W"] + p["segment1.b"]
    zn = torch.nn.functional.normalize(z, dim=1)
    ref = tr.ap_loss(torch.tensor(y.astype(np.int64)), zn, N)
    ref.backward()
    assert abs(float(loss) - float(ref.detach())) < 1e-5
    for k in p:
        rg = p[k].grad.numpy()
        assert np.abs(m.param(k, grad=True).cpu().numpy() - rg).max() <= 2e-3 * np.abs(rg).max(), k
    l0 = float(t.train_step(_dev(x), _dev(y, np.int32)))
    assert np.isfinite(l0) and 0.0 <= float(metric.result()) <= 1.0


def test_config4_mfcc_cmvn_cnn_train_step():
    """BASELINE configs[3]: MFCC(1:13) + CMVN front-end -> lidbox.models.cnn, train step from waveforms"""
    from lidbox_amd import _native as nv
    from lidbox_amd.features import audio
    from lidbox_amd.models import cnn
    from lidbox_amd.testutil import synthetic_batch
    from lidbox_amd.train import Trainer
    sig, y = synthetic_batch(6, num_labels=4, duration_s=1.0)
    sd, yd = _dev(sig), _dev(y, np.int32)
    m = cnn.create((98, 12), 4, seed=5)
    plan = audio.get_plan(16000, 400, 160)
    t = Trainer(m, feature=dict(plan=plan, kind=nv.FEAT_MFCC, cmvn=True), use_graph=True)
    # oracle: MFCC + CMVN (window_len=-1) -> cnn forward -> sparse CE
    feats = fo.extract_features(sig, [16000] * 6, "mfcc", window_norm_kwargs=dict(window_len=-1, normalize_variance=True))
    assert feats.shape == (6, 98, 12)
    ref_loss = mo.sparse_ce_from_logits(mo.cnn_fwd(_oracle_params(m), feats), y)
    l0 = float(t.train_step(sd, yd))
    assert abs(l0 - ref_loss) <= 2e-4 * abs(ref_loss), (l0, ref_loss)
    for _ in range(15):
        l1 = float(t.train_step(sd, yd))
    assert np.isfinite(l1) and l1 < l0


def test_config5_xvector_ap_cavg_100_languages():
    """BASELINE configs[4] in fp32: x-vector trunk -> segment1 (affine, D=512) -> L2 norm ->
    SparseAngularProximity(N=100) + C_avg(100 thresholds on -theta), one train step from waveforms"""
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
    convs = [xvector.frame_layer(512, 5, 1, name="frame1"), xvector.frame_layer(512, 3, 2, name="frame2"),
             xvector.frame_layer(512, 3, 3, name="frame3"), xvector.frame_layer(512, 1, 1, name="frame4"),
             xvector.frame_layer(1500, 1, 1, name="frame5")]
    m = SequentialTDNN((48, 40), convs, "stats", [DenseSpec("segment1", D, relu=False)], output_activation=None, seed=0)
    metric = SparseAverageDetectionCost(N, np.linspace(-np.pi, 0, 100))
    t = Trainer(m, loss=SparseAngularProximity(N, D), feature=dict(plan=audio.get_plan(16000, 400, 160), kind=nv.FEAT_LOGMEL),
                use_graph=True, metric=metric)
    sd, yd = _dev(sig), _dev(y, np.int32)
    # oracle loss at step 0
    feats = fo.extract_features(sig, [16000] * B, "logmelspectrogram")
    p = _oracle_params(m)
    h = feats
    for name, f, k, s_ in mo.XVECTOR_FRAMES:
        h = mo.conv1d_causal_fwd(h, p[name + ".W"], p[name + ".b"], s_)
    z = mo.dense_fwd(mo.stats_pool_fwd(h), p["segment1.W"], p["segment1.b"], relu=False)
    ref = mo.ap_loss(y, mo.l2_normalize(z), N)
    l0 = float(t.train_step(sd, yd))
    assert abs(l0 - ref) <= 1e-4 * abs(ref), (l0, ref)
    for _ in range(5):
        l1 = float(t.train_step(sd, yd))
    assert np.isfinite(l1) and l1 < l0
    c = float(metric.result())
    assert 0.0 <= c <= 1.0
    # 6 steps x B examples x 100 thresholds: each example lands in tp or fn of its label once per threshold
    # (and the pre-capture warm-up pass must NOT have been counted)
    assert float(metric.tp.sum() + metric.fn.sum()) == 6 * B * 100


def test_xvector_extended_matches_oracle_forward_and_backward():
    """reference xvector_extended.py:22-43 -- strides 1/2/3/4, incl. k=3 with stride 4"""
    from lidbox_amd.models import xvector_extended
    from lidbox_amd.train import Trainer
    rng = np.random.default_rng(21)
    B, T, C, N = 3, 130, 20, 6
    x = rng.standard_normal((B, T, C))
    y = rng.integers(0, N, size=B).astype(np.int32)
    m = xvector_extended.create((T, C), N, seed=9)
    assert [c.name for c in m.convs] == ["frame%d" % i for i in range(1, 11)]
    rngb = np.random.default_rng(22)
    m.set_weights({k: rngb.standard_normal(v.shape) * 0.05 for k, v in m.get_weights().items() if k.endswith(".b")})
    p = _oracle_params(m)
    # oracle forward / backward through the generic conv + pool + dense restatement
    acts, h = [x], x
    for (f, k, s_), c in zip(xvector_extended.FRAMES, m.convs):
        h = mo.conv1d_causal_fwd(h, p[c.name + ".W"], p[c.name + ".b"], s_)
        acts.append(h)
    assert [a.shape[1] for a in acts] == [130, 130, 130, 65, 65, 22, 22, 6, 6, 6, 6]
    pooled = mo.stats_pool_fwd(h)
    s1 = mo.dense_fwd(pooled, p["segment1.W"], p["segment1.b"])
    s2 = mo.dense_fwd(s1, p["segment2.W"], p["segment2.b"])
    z = mo.dense_fwd(s2, p["output.W"], p["output.b"], relu=False)
    logp = mo.log_softmax(z)
    got = m(_dev(x)).cpu().numpy()
    assert np.abs(got - logp).max() < 1e-3
    ref_loss = mo.sparse_ce_from_logits(logp, y)
    dz = mo.sparse_ce_from_logits_grad(logp, y)
    g = {}
    dh, g["output.W"], g["output.b"] = mo.dense_bwd(s2, p["output.W"], z, dz, relu=False)
    dh, g["segment2.W"], g["segment2.b"] = mo.dense_bwd(s1, p["segment2.W"], s2, dh)
    dh, g["segment1.W"], g["segment1.b"] = mo.dense_bwd(pooled, p["segment1.W"], s1, dh)
    dh = mo.stats_pool_bwd(acts[-1], dh)
    for i in range(9, -1, -1):
        c = m.convs[i]
        dh, g[c.name + ".W"], g[c.name + ".b"] = mo.conv1d_causal_bwd(acts[i], p[c.name + ".W"], acts[i + 1], dh,
                                                                       xvector_extended.FRAMES[i][2], need_dx=(i > 0))
    loss, _ = Trainer(m, use_graph=False).loss_and_grads(_dev(x), _dev(y, np.int32))
    assert abs(float(loss) - ref_loss) <= 1e-4 * abs(ref_loss)
    for name, ref in g.items():
        gotg = m.param(name, grad=True).cpu().numpy()
        assert np.abs(gotg - ref).max() <= 1e-3 * max(1e-12, np.abs(ref).max()), name


def test_trainer_applies_spatial_dropout_with_a_fresh_mask_per_graph_replay():
    """`Trainer` is the only training path, so the model's channel_dropout_rate must act there
    (xvector.py:50-51), also behind the fused feature kernel and under hipGraph replay."""
    from lidbox_amd import _native as nv
    from lidbox_amd.features import audio
    from lidbox_amd.models import xvector
    from lidbox_amd.testutil import synthetic_batch
    from lidbox_amd.train import Trainer
    sig, y = synthetic_batch(8, num_labels=4, duration_s=0.5)
    sd, yd = _dev(sig), _dev(y, np.int32)
    plan = audio.get_plan(16000, 400, 160)
    for use_graph in (False, True):
        m = xvector.create((48, 40), 4, channel_dropout_rate=0.5, seed=3)
        t = Trainer(m, feature=dict(plan=plan, kind=nv.FEAT_LOGMEL), use_graph=use_graph)
        ref = Trainer(xvector.create((48, 40), 4, seed=3), feature=dict(plan=plan, kind=nv.FEAT_LOGMEL), use_graph=False)
        ref.loss_and_grads(sd, yd)
        clean = ref.model.workspace(8, 48).input_view().clone()          # log-mel features without dropout
        seen = []
        for _ in range(3):
            t.train_step(sd, yd)
            x = m.workspace(8, 48).input_view()
            dropped = (x == 0).all(dim=1)                                # [B, C]: channel zero for every frame
            kept = ~dropped
            assert 0.25 < float(dropped.float().mean()) < 0.75
            # kept channels hold 2 x the clean features (1 / (1 - 0.5)), dropped ones exact zeros
            scale = torch.where(kept, 2.0, 0.0)[:, None, :]
            assert float((x - clean * scale).abs().max()) <= 1e-5 * float(clean.abs().max())
            seen.append(dropped.cpu().numpy())
        assert not np.array_equal(seen[0], seen[1]) and not np.array_equal(seen[1], seen[2])
    # feature-tensor inputs take the same path
    m = xvector.create((48, 40), 4, channel_dropout_rate=0.5, seed=3)
    t = Trainer(m, use_graph=True)
    t.train_step(clean.contiguous(), yd)
    x = m.workspace(8, 48).input_view()
    assert 0.25 < float((x == 0).all(dim=1).float().mean()) < 0.75
    # inference never drops
    assert float((m(clean.contiguous()) - m(clean.contiguous())).abs().max()) == 0.0


def test_config0_32_utterances_logmel_xvector_forward():
    """BASELINE configs[0]: 32 x (16 kHz, 2 s) synthetic wavs -> log-mel -> x-vector forward, against the oracle"""
    from lidbox_amd.data import tf_utils
    from lidbox_amd.models import xvector
    from lidbox_amd.testutil import synthetic_batch
    sig, _ = synthetic_batch(32, num_labels=4, duration_s=2.0)
    m = xvector.create((198, 40), 4, seed=0)
    feats = tf_utils.extract_features(_dev(sig), [16000] * 32, "logmelspectrogram")
    ref_feats = fo.extract_features(sig, [16000] * 32, "logmelspectrogram")
    got_feats = feats.cpu().numpy()
    assert got_feats.shape == (32, 198, 40)
    assert np.abs(got_feats - ref_feats).max() < 2e-4
    got = m(feats).cpu().numpy()
    ref = mo.xvector_fwd(_oracle_params(m), ref_feats)
    assert got.shape == (32, 4)
    assert np.abs(got - ref).max() < 1e-3
    emb = xvector.as_embedding_extractor(m)(feats).cpu().numpy()
    assert _cos(emb, mo.xvector_fwd(_oracle_params(m), ref_feats, embedding=True)).min() >= 0.9999


def test_learning_rate_schedule_reaches_the_captured_optimizer():
    """reference keras_utils.py:137-139 (`lr_scheduler`): the schedule's value of every step is written into the device-side
    Adam state ahead of the step, so eager and graph-replayed steps follow it -- against the oracle's Adam run with the
    same per-step learning rates, and a piecewise schedule that drops to (almost) zero freezes the weights"""
    from lidbox_amd.models import xvector
    from lidbox_amd.models.keras_utils import lr_schedule_from_config
    from lidbox_amd.train import Trainer
    rng = np.random.default_rng(11)
    x = rng.standard_normal((6, 50, 24)).astype(np.float32)
    y = rng.integers(0, 5, size=6).astype(np.int32)
    xd, yd = _dev(x), _dev(y, np.int32)
    sched = lr_schedule_from_config({"cls": "ExponentialDecay",
                                     "kwargs": {"initial_learning_rate": 2e-3, "decay_steps": 2, "decay_rate": 0.5}})
    m_e, m_g = xvector.create((50, 24), 5, seed=7), xvector.create((50, 24), 5, seed=7)
    p = _oracle_params(m_e)
    am = {k: np.zeros_like(v) for k, v in p.items()}
    av = {k: np.zeros_like(v) for k, v in p.items()}
    t_e = Trainer(m_e, use_graph=False, optimizer=dict(lr_schedule=sched))
    t_g = Trainer(m_g, use_graph=True, optimizer=dict(lr_schedule=sched))
    for step in range(1, 5):
        le, lg = float(t_e.train_step(xd, yd)), float(t_g.train_step(xd, yd))
        lo, go, _ = mo.xvector_loss_and_grads(p, x.astype(np.float64), y)
        mo.adam_step(p, go, am, av, step, lr=sched(step - 1))
        assert le == lg and abs(le - lo) <= 2e-4 * abs(lo), (step, le, lg, lo)
    assert torch.equal(m_e.flat, m_g.flat)
    for k, v in m_e.get_weights().items():
        # Adam normalises the update: a weight whose gradient is ~0 may take its +-lr step the other way on fp32 noise;
        # four steps at <= 2e-3 moved the weights by ~6e-3
        assert np.abs(v - p[k]).max() <= 1e-3, k
        assert np.median(np.abs(v - p[k])) <= 2e-5, k
    # piecewise: one real step, then a vanishing rate
    pw = lr_schedule_from_config({"cls": "PiecewiseConstantDecay", "kwargs": {"boundaries": [0], "values": [1e-3, 1e-12]}})
    m = xvector.create((50, 24), 5, seed=7)
    t = Trainer(m, use_graph=True, optimizer=dict(lr_schedule=pw))
    w0 = m.flat.clone()
    t.train_step(xd, yd)
    w1 = m.flat.clone()
    for _ in range(3):
        t.train_step(xd, yd)
    assert float((w1 - w0).abs().max()) > 5e-4 and float((m.flat - w1).abs().max()) < 1e-9
    # a schedule that returns EXACTLY 0 freezes the weights bit for bit (a zero rate must not fall back to the
    # constant `lr` argument -- the host writes it as -0.0, an all-zero word means "no schedule")
    pz = lr_schedule_from_config({"cls": "PiecewiseConstantDecay", "kwargs": {"boundaries": [0], "values": [1e-3, 0.0]}})
    for use_graph in (False, True):
        m = xvector.create((50, 24), 5, seed=7)
        t = Trainer(m, use_graph=use_graph, optimizer=dict(lr_schedule=pz))
        t.train_step(xd, yd)
        w1 = m.flat.clone()
        for _ in range(2):
            t.train_step(xd, yd)
        assert torch.equal(m.flat, w1) and t.step_count == 3
    # the schedule follows the DEVICE step when the optimizer state is re-pointed (bench.py shares one state between two trainers)
    m = xvector.create((50, 24), 5, seed=7)
    t1 = Trainer(m, use_graph=False, optimizer=dict(lr_schedule=pz))
    t1.train_step(xd, yd)
    t2 = Trainer(m, use_graph=False, optimizer=dict(lr_schedule=pz))
    t2.m, t2.v, t2.adam_state = t1.m, t1.v, t1.adam_state
    w1 = m.flat.clone()
    t2.train_step(xd, yd)                      # device step 1 -> the schedule's second value (0), not its first
    assert torch.equal(m.flat, w1)


@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
def test_jobs_left_by_an_interrupted_pass_are_dropped(dtype):
    """A backward pass that raised half-way leaves carried-reduce jobs in the workspace's list.  The next pass must not run
    them: here the left-over is a zero-fill job over frame5's weight gradient, which would wipe that layer's update.  Two
    eager steps with the injection equal two steps of an untouched twin bit for bit."""
    import ctypes
    from lidbox_amd import _native as nv
    from lidbox_amd.features import audio
    from lidbox_amd.models import xvector
    from lidbox_amd.testutil import synthetic_batch
    from lidbox_amd.train import Trainer
    sig, y = synthetic_batch(6, num_labels=4, duration_s=1.0, seed=11)
    x, lab = _dev(sig), _dev(y, np.int32)
    plan = audio.get_plan(16000, 400, 160)
    ms = [xvector.create((98, 40), 4, seed=0, compute_dtype=dtype) for _ in range(2)]
    ts = [Trainer(m, feature=dict(plan=plan, kind=nv.FEAT_LOGMEL), use_graph=False) for m in ms]
    for t in ts:
        t.train_step(x, lab)
    ws = ms[0].workspace(6, 98)
    off, shape = ms[0].layout["frame5.W"]
    n = int(np.prod(shape))
    stale = nv.ReduceJob()
    nv.check(nv.lib.lidbox_zero_job(ctypes.c_void_p(ms[0].flat_grad.data_ptr() + 4 * off), n, n, 1, ctypes.byref(stale)))
    ws.pending.append((stale, -1))
    la, lb = float(ts[0].train_step(x, lab)), float(ts[1].train_step(x, lab))
    assert la == lb and torch.equal(ms[0].flat, ms[1].flat)
    assert not ws.pending


@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
def test_train_step_runs_one_standalone_slice_sum(dtype, monkeypatch):
    """The wgrad slice sums of the x-vector step ride in the leading workgroups of later GEMM launches (carried reduces):
    at the benchmark's batch size exactly ONE launch of their own is left per step -- frame1's (no dgrad behind it), together
    with the optimizer's scalar job -- and the conv-layer dgrad calls carry the others."""
    from lidbox_amd import _native as nv
    from lidbox_amd.features import audio
    from lidbox_amd.models import xvector
    from lidbox_amd.testutil import synthetic_batch
    from lidbox_amd.train import Trainer
    sig, y = synthetic_batch(256, num_labels=4, duration_s=2.0, seed=5)     # BASELINE configs[1]'s per-GPU batch: the decompositions are per shape
    x, lab = _dev(sig), _dev(y, np.int32)
    m = xvector.create((198, 40), 4, seed=0, compute_dtype=dtype)
    tr = Trainer(m, feature=dict(plan=audio.get_plan(16000, 400, 160), kind=nv.FEAT_LOGMEL), use_graph=False)
    tr.train_step(x, lab)                                       # allocate, warm up
    calls = {"standalone": 0, "jobs_in_standalone": 0, "carriers": 0, "carried": 0}
    real_run = nv.lib.lidbox_reduce_jobs_run
    carry_names = ["lidbox_gemm_nt_carry", "lidbox_gemm_nt_tn_carry"] if dtype == "float32" else ["lidbox_gemm_bf16s_nt_carry", "lidbox_gemm_nt_tn_carry"]
    last = {"lidbox_gemm_nt_carry": nv.lib.lidbox_gemm_last_carried, "lidbox_gemm_nt_tn_carry": nv.lib.lidbox_gemm_last_carried,
            "lidbox_gemm_bf16s_nt_carry": nv.lib.lidbox_gemm_bf16s_last_carried}

    def run(jobs, n, stream):
        calls["standalone"] += 1
        calls["jobs_in_standalone"] += n
        return real_run(jobs, n, stream)
    monkeypatch.setattr(nv.lib, "lidbox_reduce_jobs_run", run)
    for name in carry_names:
        real = getattr(nv.lib, name)

        def wrapped(*a, _real=real, _name=name):
            rc = _real(*a)
            calls["carriers"] += 1
            calls["carried"] += int(last[_name]() > 0)
            return rc
        monkeypatch.setattr(nv.lib, name, wrapped)
    loss = float(tr.train_step(x, lab))
    assert np.isfinite(loss)
    assert calls["standalone"] == 1 and calls["jobs_in_standalone"] == 2, calls       # frame1's slices + the optimizer's scalars
    assert calls["carriers"] >= 5 and calls["carried"] >= 4, calls
