"""
GPU parity of BatchNormalization (csrc/batchnorm.hip) and the xvector_2d model (reference lidbox/models/xvector_2d.py)
against the numpy oracle and torch autograd on the CPU restatement.  PARITY UNPINNED like every model row (no TensorFlow
here): Keras BatchNormalization / Conv2D semantics are restated from their documented behaviour.
"""
import ctypes

import numpy as np
import pytest
import torch

from oracle import model_np as mo
from oracle import torch_ref as tr

pytestmark = pytest.mark.gpu


def _dev(x, dtype=np.float32):
    return torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype))).cuda()


@pytest.mark.parametrize("R,C", [(1000, 64), (37, 5), (4097, 256), (1, 3), (20000, 32)])
def test_batchnorm_kernels_forward_backward(R, C):
    from lidbox_amd import _native as nv
    rng = np.random.default_rng(R + C)
    st = nv.current_stream()
    x = np.maximum(rng.standard_normal((R, C)) * rng.uniform(0.1, 3, C) + rng.uniform(-1, 5, C), 0)     # ReLU outputs
    gamma, beta = rng.uniform(0.5, 1.5, C), rng.standard_normal(C)
    mm0, mv0 = rng.standard_normal(C), rng.uniform(0.5, 2, C)
    xd, gd, bd = _dev(x), _dev(gamma), _dev(beta)
    mm, mv = _dev(mm0), _dev(mv0)
    consts = torch.zeros((4, C), device="cuda")
    cp = [ctypes.c_void_p(consts.data_ptr() + 4 * j * C) for j in range(4)]
    wsb = int(nv.lib.lidbox_bn_workspace(R, C))
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    nv.check(nv.lib.lidbox_bn_train_stats(nv.ptr(xd), R, C, nv.ptr(gd), nv.ptr(bd), 1e-3, 0.99, nv.ptr(mm), nv.ptr(mv),
                                          cp[0], cp[1], cp[2], cp[3], nv.ptr(ws), wsb, st))
    x32 = x.astype(np.float32).astype(np.float64)
    ref_y, ref_mm, ref_mv = mo.batchnorm_fwd(x32, gamma, beta, mm0, mv0, True)
    mean, var = x32.mean(0), x32.var(0)
    assert np.abs(consts[0].cpu().numpy() - mean).max() <= 1e-6 * max(1.0, np.abs(mean).max())
    assert np.abs(consts[1].cpu().numpy() - 1 / np.sqrt(var + 1e-3)).max() <= 2e-6 * (1 / np.sqrt(var + 1e-3)).max()
    assert np.abs(mm.cpu().numpy() - ref_mm).max() < 1e-6 and np.abs(mv.cpu().numpy() - ref_mv).max() < 1e-5
    # apply: dense and through a strided rows descriptor (rows of 2 land in a padded [R/2.., 3 + 2, C] buffer)
    y = torch.zeros((R, C), device="cuda")
    nv.check(nv.lib.lidbox_bn_apply(nv.ptr(xd), R, C, cp[2], cp[3], nv.Rows(y.data_ptr(), 0, C, 1, R), st))
    assert np.abs(y.cpu().numpy() - ref_y).max() <= 2e-5 * max(1.0, np.abs(ref_y).max())
    if R % 2 == 0:
        padded = torch.zeros((R // 2, 5, C), device="cuda")
        nv.check(nv.lib.lidbox_bn_apply(nv.ptr(xd), R, C, cp[2], cp[3], nv.Rows(padded.data_ptr() + 4 * 3 * C, 5 * C, C, R // 2, 2), st))
        assert torch.equal(padded[:, 3:].reshape(R, C), y) and not bool(padded[:, :3].any())
    # inference constants
    nv.check(nv.lib.lidbox_bn_infer_consts(nv.ptr(gd), nv.ptr(bd), nv.ptr(mm), nv.ptr(mv), 1e-3, C, cp[2], cp[3], st))
    nv.check(nv.lib.lidbox_bn_apply(nv.ptr(xd), R, C, cp[2], cp[3], nv.Rows(y.data_ptr(), 0, C, 1, R), st))
    ref_inf, _, _ = mo.batchnorm_fwd(x32, gamma, beta, mm.cpu().numpy().astype(np.float64), mv.cpu().numpy().astype(np.float64), False)
    assert np.abs(y.cpu().numpy() - ref_inf).max() <= 2e-5 * max(1.0, np.abs(ref_inf).max())
    # backward vs torch autograd (float64) of gamma * (relu(z) - mean) / sqrt(var + eps) + beta at z = x (x > 0 where it counts)
    dy = rng.standard_normal((R, C))
    zt = torch.tensor(x32, requires_grad=True)
    gt, bt = torch.tensor(gamma, requires_grad=True), torch.tensor(beta, requires_grad=True)
    at = torch.relu(zt)
    yt = gt * (at - at.mean(0)) / torch.sqrt(at.var(0, unbiased=False) + 1e-3) + bt
    (yt * torch.tensor(dy)).sum().backward()
    dgam, dbet, dx = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda"), torch.zeros((R, C), device="cuda")
    dyd = _dev(dy)
    nv.check(nv.lib.lidbox_bn_bwd(nv.ptr(xd), nv.Rows(dyd.data_ptr(), 0, C, 1, R), R, C, cp[0], cp[1], nv.ptr(gd), 1,
                                  nv.ptr(dgam), nv.ptr(dbet), nv.ptr(dx), nv.ptr(ws), wsb, st))
    for got, ref in ((dgam, gt.grad), (dbet, bt.grad), (dx, zt.grad)):
        ref = ref.numpy()
        assert np.abs(got.cpu().numpy() - ref).max() <= 1e-4 * max(1e-6, np.abs(ref).max())


def _oracle_params(model):
    return {k: v.astype(np.float64) for k, v in model.get_weights().items()}


def test_xvector_2d_layout_forward_and_moving_statistics():
    from lidbox_amd.models import xvector_2d
    m = xvector_2d.create((30, 40), 4, seed=2)
    ref_p = mo.xvector_2d_init(40, 4, seed=0)
    assert {k: v.shape for k, v in m.get_weights().items()} == {k: v.shape for k, v in ref_p.items()}
    assert m.count_params() == sum(v.size for v in ref_p.values()) and m.input_dim == 32 and m.model_input_dim == 40
    rng = np.random.default_rng(3)
    w = m.get_weights()
    for k in w:                                           # non-trivial BatchNorm parameters and running statistics
        if k.endswith("gamma"): w[k] = rng.uniform(0.5, 1.5, w[k].shape)
        if k.endswith("beta") or k.endswith("_conv.b"): w[k] = rng.standard_normal(w[k].shape) * 0.1
        if k.endswith("moving_mean"): w[k] = rng.uniform(0, 0.5, w[k].shape)
        if k.endswith("moving_variance"): w[k] = rng.uniform(0.5, 1.5, w[k].shape)
    m.set_weights(w)
    p = _oracle_params(m)
    x = rng.standard_normal((3, 30, 40))
    # inference: running statistics, nothing moves
    ref, _ = mo.xvector_2d_fwd(p, x, training=False)
    got = m(_dev(x), training=False).cpu().numpy()
    assert got.shape == (3, 4) and np.abs(got - ref).max() < 1e-3
    assert all(np.array_equal(m.get_weights()[k], w[k].astype(np.float32)) for k in w if "moving" in k)
    # training: batch statistics, running statistics move by (1 - 0.99) towards them
    ref, new_stats = mo.xvector_2d_fwd(p, x, training=True)
    got = m(_dev(x), training=True).cpu().numpy()
    assert np.abs(got - ref).max() < 1e-3
    after = m.get_weights()
    for k, v in new_stats.items():
        assert np.abs(after[k] - v).max() <= 1e-5 * max(1.0, np.abs(v).max()), k
        assert not np.array_equal(after[k], w[k].astype(np.float32))
    # embedding extractor: segment1's affine output with the running statistics
    emb = xvector_2d.as_embedding_extractor(m)(_dev(x)).cpu().numpy()
    ref_emb, _ = mo.xvector_2d_fwd(_oracle_params(m), x, training=False, embedding=True)
    cos = (emb * ref_emb).sum(-1) / (np.linalg.norm(emb, axis=-1) * np.linalg.norm(ref_emb, axis=-1))
    assert emb.shape == (3, 512) and cos.min() >= 0.9999
    with pytest.raises(ValueError):
        xvector_2d.create((30, 12), 4)                   # 12 channels do not survive the four valid convolutions


@pytest.mark.parametrize("F", [40, 64])
def test_xvector_2d_gradients_match_autograd_and_training_learns(F):
    """F = 64 leaves cols = 3 positions of 32 channels after the front-end (flatten_channels, dgrad into strided rows)"""
    from lidbox_amd.models import xvector_2d
    from lidbox_amd.train import Trainer
    rng = np.random.default_rng(5)
    B, T = 4, 21
    m = xvector_2d.create((T, F), 3, seed=7)
    assert m.fe_dims[-1] == (1 if F == 40 else 3)
    w = m.get_weights()
    for k in w:
        if k.endswith("gamma"): w[k] = rng.uniform(0.5, 1.5, w[k].shape)
        if k.endswith("beta") or k.endswith(".b"): w[k] = rng.standard_normal(w[k].shape) * 0.1
    m.set_weights(w)
    x = rng.standard_normal((B, T, F))
    y = rng.integers(0, 3, size=B).astype(np.int32)
    p = _oracle_params(m)
    pt = tr.to_torch_params({k: v for k, v in p.items()}, True, torch.float64)
    loss_ref = tr.sparse_ce_from_logits(tr.xvector_2d_fwd(pt, torch.tensor(x), training=True), torch.tensor(y.astype(np.int64)))
    loss_ref.backward()
    t = Trainer(m, use_graph=False)
    before = {k: v.copy() for k, v in m.get_weights().items() if "moving" in k}
    loss, _ = t.loss_and_grads(_dev(x), _dev(y, np.int32))
    assert abs(float(loss) - float(loss_ref.detach())) <= 1e-4 * abs(float(loss_ref.detach()))
    flips = 0
    for k in m.layout:
        ref_g = pt[k].grad.numpy()
        got_g = m.param(k, grad=True).cpu().numpy()
        d, scale = np.abs(got_g - ref_g), max(1e-12, np.abs(ref_g).max())
        if d.max() > 2e-3 * scale:
            # One ReLU whose pre-activation lies within fp32 round-off of zero may fall on the other side than in float64
            # (311 k activations in frame2d_2 alone; which one depends on the GEMM's summation order, i.e. on the
            # planner): the error is then confined to ONE output channel of one layer (tools/scratch/x2d_flip.py at commit 0ed6482:
            # channel 71 at 4e-3, every other channel at 1e-6) -- allowed once per model, everything else is an error
            per_c = d.reshape(-1, d.shape[-1]).max(0)
            worst = int(np.argmax(per_c))
            assert np.delete(per_c, worst).max() <= 2e-5 * scale and per_c[worst] <= 2e-2 * scale, k
            flips += 1
    assert flips <= 2          # a layer's W and b
    # the captured step: running statistics move once per step (not in the warm-up pass), the loss goes down
    m2 = xvector_2d.create((T, F), 3, seed=7)
    t2 = Trainer(m2, use_graph=True)
    xd, yd = _dev(x), _dev(y, np.int32)
    l0 = float(t2.train_step(xd, yd))
    s1 = m2.get_weights()["frame2d_1_bn.moving_variance"].copy()
    _, stats = mo.xvector_2d_fwd({k: v.astype(np.float64) for k, v in xvector_2d.create((T, F), 3, seed=7).get_weights().items()}, x, training=True)
    assert np.abs(s1 - stats["frame2d_1_bn.moving_variance"]).max() < 1e-5          # exactly one update after one step
    for _ in range(10):
        l1 = float(t2.train_step(xd, yd))
    assert np.isfinite(l1) and l1 < l0
    assert len(before) == 8


def test_xvector_2d_train_step_from_waveforms_and_bf16_compute():
    """the fused log-mel kernel writes straight into the 2-D front-end's input buffer; the bf16 GEMM family runs the
    front-end's deeper layers and the TDNN (layer 1, K = 5 single-channel taps, always stays in the fp32 family)"""
    from lidbox_amd import _native as nv
    from lidbox_amd.features import audio
    from lidbox_amd.models import xvector_2d
    from lidbox_amd.testutil import synthetic_batch
    from lidbox_amd.train import Trainer
    from oracle import features_np as fo
    sig, y = synthetic_batch(6, num_labels=4, duration_s=0.5)
    sd, yd = _dev(sig), _dev(y, np.int32)
    plan = audio.get_plan(16000, 400, 160)
    m = xvector_2d.create((48, 40), 4, seed=4)
    t = Trainer(m, feature=dict(plan=plan, kind=nv.FEAT_LOGMEL), use_graph=True)
    feats = fo.extract_features(sig, [16000] * 6, "logmelspectrogram")
    ref, _ = mo.xvector_2d_fwd(_oracle_params(m), feats, training=True)
    ref_loss = mo.sparse_ce_from_logits(ref, y)
    l0 = float(t.train_step(sd, yd))
    assert abs(l0 - ref_loss) <= 2e-4 * abs(ref_loss), (l0, ref_loss)
    assert np.abs(m.workspace(6, 48).input_view().cpu().numpy() - feats).max() < 1e-3
    for _ in range(8):
        l1 = float(t.train_step(sd, yd))
    assert np.isfinite(l1) and l1 < l0
    # bf16 compute: same model, operands rounded on chip; tracks the fp32 loss at bf16 accuracy
    mb = xvector_2d.create((48, 40), 4, seed=4, compute_dtype="bfloat16")
    assert not mb.bf16_storage                       # the shadow path is the plain TDNN's; the front-end model keeps fp32-source kernels
    lb, _ = Trainer(mb, feature=dict(plan=plan, kind=nv.FEAT_LOGMEL), use_graph=False).loss_and_grads(sd, yd)
    assert abs(float(lb) - ref_loss) <= 3e-2 * abs(ref_loss)


def test_frame_layer_2d_dropout_forward_mask_and_backward():
    """FrameLayer2D(dropout_rate=...) (reference xvector_2d.py:37-39,45-46: Keras Dropout behind the BatchNormalization;
    the reference's create() never sets it).  Inference ignores it; in training an element is zeroed with probability
    `rate` and the kept ones are the BatchNormalization output / (1 - rate); the mask depends on (seed, optimizer step,
    position) only -- the same within a step, new after an optimizer step -- and the backward pass applies the same mask:
    the analytic gradient equals a central finite difference of the (deterministic at a fixed step) loss."""
    from lidbox_amd.models import xvector_2d
    from lidbox_amd.models.tdnn import DenseSpec, SequentialTDNN
    from lidbox_amd.models.xvector import frame_layer, segment_layer
    from lidbox_amd.train import Trainer
    rate = 0.3

    def build(rate_):
        frontend = [xvector_2d.FrameLayer2D(16, (1, 5), (1, 2), name="frame2d_1", dropout_rate=rate_),
                    xvector_2d.FrameLayer2D(8, (1, 3), (1, 3), name="frame2d_2")]
        convs = [frame_layer(32, 3, 1, name="frame1"), frame_layer(48, 1, 1, name="frame2")]
        denses = [segment_layer(24, name="segment1"), DenseSpec("output", 3, relu=False)]
        return SequentialTDNN((12, 40), convs, "stats", denses, name="tiny-2d", output_activation="log_softmax", seed=4,
                              frontend=frontend)
    rng = np.random.default_rng(8)
    x = _dev(rng.standard_normal((5, 12, 40)))
    y = _dev(rng.integers(0, 3, size=5), np.int32)
    m, m0 = build(rate), build(0.0)
    assert torch.equal(m.flat, m0.flat)
    # inference: no dropout
    assert torch.equal(m(x, training=False), m0(x, training=False))
    # training forward: layer 1's output (input rows of layer 2) is the no-dropout output times {0, 1/(1-rate)}
    m0(x, training=True)
    clean = m0.workspace(5, 12).fe_y[0].clone()
    m(x, training=True)
    dropped = m.workspace(5, 12).fe_y[0].clone()
    zero = dropped == 0
    frac = float((zero & (clean != 0)).float().sum() / (clean != 0).float().sum())
    assert abs(frac - rate) < 0.03, frac
    assert torch.equal(dropped[~zero], (clean * (1.0 / (1.0 - rate)))[~zero])
    m(x, training=True)
    assert torch.equal(m.workspace(5, 12).fe_y[0], dropped)              # same step -> same mask
    # backward: analytic gradient vs central differences along a random direction of the first conv's weights
    tr = Trainer(m, use_graph=False)
    loss, g = tr.loss_and_grads(x, y)
    g = g.clone()
    off, shape = m.layout["frame2d_1_conv.W"]
    n = int(np.prod(shape))
    d = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).cuda()
    d /= d.norm()
    eps = 2e-3
    base = m.flat.clone()
    vals = []
    for sgn in (1.0, -1.0):
        m.flat.copy_(base)
        m.flat[off:off + n] += sgn * eps * d
        vals.append(float(tr.loss_and_grads(x, y)[0]))
    m.flat.copy_(base)
    fd = (vals[0] - vals[1]) / (2 * eps)
    an = float((g[off:off + n] * d).sum())
    assert abs(fd - an) <= 0.03 * max(abs(an), 1e-3) + 2e-4, (fd, an)
    # a new optimizer step draws a new mask
    tr.train_step(x, y)
    m(x, training=True)
    assert not torch.equal(m.workspace(5, 12).fe_y[0] == 0, zero)
