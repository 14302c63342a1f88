"""
GPU parity of the SURVEY 8f.1 x-vector variants against the oracle: the frequency-attention x-vector
(reference xvector_freq_attention.py:22-43, clstm.py:31-42), the attention kernels on their own, and the
opt-in dilated causal Conv1D (Keras dilation_rate semantics).

Tolerances as in test_model_gpu.py: log-prob max-abs 1e-3, loss rel 1e-4, per-layer gradient rel 1e-3,
embedding cosine >= 0.9999; bare kernels rel 2e-5.
"""
import numpy as np
import pytest
import torch

from oracle import model_np as mo

pytestmark = pytest.mark.gpu


def _dev(x, dtype=np.float32):
    return torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype))).cuda()


def _oracle_params(model):
    return {k: v.astype(np.float64) for k, v in model.get_weights().items()}


def _cos(a, b):
    return (a * b).sum(-1) / (np.linalg.norm(a, axis=-1) * np.linalg.norm(b, axis=-1))


@pytest.mark.parametrize("rows,C,d_f,relu", [(37, 1500, 60, 1), (5, 64, 16, 0), (130, 96, 3, 1), (1, 7, 7, 1),
                                             (9, 2048, 64, 1), (3, 4000, 8, 0)])
def test_freq_attention_kernels(rows, C, d_f, relu):
    from lidbox_amd import _native as nv
    rng = np.random.default_rng(rows + C)
    H = np.maximum(rng.standard_normal((rows, C)), 0)
    logits = rng.standard_normal((rows, d_f)) * 2
    F = mo.softmax(logits)
    Hw = (H.reshape(rows, d_f, C // d_f) * F[..., None]).reshape(rows, C)
    h, lg = _dev(H), _dev(logits)
    hw = torch.empty_like(h)
    st = nv.current_stream()
    nv.check(nv.lib.lidbox_freq_attention_fwd(nv.ptr(h), nv.ptr(lg), rows, C, d_f, nv.ptr(lg), nv.ptr(hw), st))
    assert np.abs(lg.cpu().numpy() - F).max() <= 2e-6
    assert np.abs(hw.cpu().numpy() - Hw).max() <= 2e-5 * np.abs(Hw).max()
    dHw = rng.standard_normal((rows, C))
    dF = (dHw.reshape(rows, d_f, -1) * H.reshape(rows, d_f, -1)).sum(-1)
    dlogit = F * (dF - (F * dF).sum(-1, keepdims=True))
    dH = (dHw.reshape(rows, d_f, -1) * F[..., None]).reshape(rows, C)
    if relu:
        dH = dH * (H > 0)
    dl, dh = torch.empty_like(lg), torch.full_like(h, 7.0)
    nv.check(nv.lib.lidbox_freq_attention_bwd(nv.ptr(h), nv.ptr(lg), nv.ptr(_dev(dHw)), rows, C, d_f, relu,
                                              nv.ptr(dl), nv.ptr(dh), st))
    assert np.abs(dl.cpu().numpy() - dlogit).max() <= 2e-5 * max(1e-9, np.abs(dlogit).max())
    assert np.abs(dh.cpu().numpy() - dH).max() <= 2e-5 * np.abs(dH).max()


def test_freq_attention_rejects_bad_bins():
    from lidbox_amd import _native as nv
    x = torch.zeros(4, 10, device="cuda")
    with pytest.raises(ValueError):
        nv.check(nv.lib.lidbox_freq_attention_fwd(nv.ptr(x), nv.ptr(x), 4, 10, 3, nv.ptr(x), nv.ptr(x), nv.current_stream()))
    from lidbox_amd.models import xvector_freq_attention
    with pytest.raises(ValueError):                      # clstm.py:32
        xvector_freq_attention.create((50, 20), 3, freq_attention_bins=7)


def _xvfa_oracle(p, x, names):
    acts, h = [x], x
    for name, (f, k, s) in zip(names, [(512, 5, 1), (512, 3, 2), (512, 3, 3), (512, 1, 1), (1500, 1, 1)]):
        h = mo.conv1d_causal_fwd(h, p[name + ".W"], p[name + ".b"], s)
        acts.append(h)
    hw, cache = mo.freq_attention_fwd(h, p["Wf_1.W"], p["Wf_2.W"], return_cache=True)
    pooled = mo.stats_pool_fwd(hw)
    return acts, hw, cache, pooled


def test_xvector_freq_attention_matches_oracle_forward_and_backward():
    """reference xvector_freq_attention.py:22-43 with the default 60 bins of 25 channels"""
    from lidbox_amd.models import xvector_freq_attention as xfa
    from lidbox_amd.train import Trainer
    rng = np.random.default_rng(31)
    B, T, C, N = 3, 61, 20, 5
    x = rng.standard_normal((B, T, C))
    y = rng.integers(0, N, size=B).astype(np.int32)
    m = xfa.create((T, C), N, seed=4)
    w = m.get_weights()
    assert w["Wf_1.W"].shape == (1500, 64) and w["Wf_2.W"].shape == (64, 60)
    assert m.count_params() == 4510176 - 2052 + 512 * N + N + 1500 * 64 + 64 * 60 + (C - 40) * 5 * 512       # x-vector + Wf_1 + Wf_2
    assert "Wf_1.b" not in w                                    # use_bias=False, clstm.py:35-36
    rb = np.random.default_rng(32)
    m.set_weights({k: rb.standard_normal(v.shape) * 0.05 for k, v in w.items() if k.endswith(".b")})
    p = _oracle_params(m)
    names = [c.name for c in m.convs]
    acts, hw, cache, pooled = _xvfa_oracle(p, x, names)
    s1 = mo.dense_fwd(pooled, p["segment1.W"], p["segment1.b"])
    s2 = mo.dense_fwd(s1, p["segment2.W"], p["segment2.b"])
    z = mo.dense_fwd(s2, p["output.W"], p["output.b"], relu=False)
    logp = mo.log_softmax(z)
    got = m(_dev(x)).cpu().numpy()
    assert np.abs(got - logp).max() < 1e-3
    emb_ref = mo.dense_fwd(pooled, p["segment1.W"], p["segment1.b"], relu=False)
    emb = xfa.as_embedding_extractor(m)(_dev(x)).cpu().numpy()
    assert _cos(emb, emb_ref).min() >= 0.9999
    # backward
    ref_loss = mo.sparse_ce_from_logits(logp, y)
    dz = mo.sparse_ce_from_logits_grad(logp, y)
    g = {}
    dh, g["output.W"], g["output.b"] = mo.dense_bwd(s2, p["output.W"], z, dz, relu=False)
    dh, g["segment2.W"], g["segment2.b"] = mo.dense_bwd(s1, p["segment2.W"], s2, dh)
    dh, g["segment1.W"], g["segment1.b"] = mo.dense_bwd(pooled, p["segment1.W"], s1, dh)
    dhw = mo.stats_pool_bwd(hw, dh)
    dh, g["Wf_1.W"], g["Wf_2.W"] = mo.freq_attention_bwd(acts[-1], p["Wf_1.W"], p["Wf_2.W"], cache, dhw)
    strides = [1, 2, 3, 1, 1]
    for i in range(4, -1, -1):
        dh, g[names[i] + ".W"], g[names[i] + ".b"] = mo.conv1d_causal_bwd(acts[i], p[names[i] + ".W"], acts[i + 1], dh,
                                                                           strides[i], need_dx=(i > 0))
    for use_graph in (False, True):
        tr = Trainer(m, use_graph=use_graph)
        loss, _ = tr.loss_and_grads(_dev(x), _dev(y, np.int32))
        assert abs(float(loss) - ref_loss) <= 1e-4 * abs(ref_loss)
        for name, ref in g.items():
            gotg = m.param(name, grad=True).cpu().numpy()
            assert np.abs(gotg - ref).max() <= 1e-3 * max(1e-12, np.abs(ref).max()), name
    # a captured train step runs and reduces the loss
    tr = Trainer(m, use_graph=True)
    xd, yd = _dev(x), _dev(y, np.int32)
    l0 = float(tr.train_step(xd, yd))
    for _ in range(5):
        l1 = float(tr.train_step(xd, yd))
    assert np.isfinite(l1) and l1 < l0


@pytest.mark.parametrize("d", [2, 3])
def test_dilated_causal_conv_matches_oracle(d):
    """Keras Conv1D(padding="causal", dilation_rate=d): forward, wgrad, bias grad and dgrad through a ReLU layer"""
    from lidbox_amd.models.tdnn import DenseSpec, SequentialTDNN
    from lidbox_amd.models.xvector import frame_layer
    from lidbox_amd.train import Trainer
    rng = np.random.default_rng(40 + d)
    B, T, C, N = 4, 37, 12, 3
    specs = [(24, 3, 1, 1), (32, 3, 1, d), (16, 2, 1, d + 1), (20, 1, 1, 5)]       # (filters, k, s, dilation)
    convs = [frame_layer(f, k, s, name="tdnn%d" % (i + 1), dilation_rate=dd) for i, (f, k, s, dd) in enumerate(specs)]
    m = SequentialTDNN((T, C), convs, "stats", [DenseSpec("fc", 8), DenseSpec("out", N, relu=False)], seed=3)
    with pytest.raises(ValueError):
        frame_layer(8, 3, 2, dilation_rate=2)
    rb = np.random.default_rng(41)
    m.set_weights({k: rb.standard_normal(v.shape) * 0.1 for k, v in m.get_weights().items() if k.endswith(".b")})
    p = _oracle_params(m)
    x = rng.standard_normal((B, T, C))
    y = rng.integers(0, N, size=B).astype(np.int32)
    acts, h = [x], x
    for c, (f, k, s, dd) in zip(m.convs, specs):
        h = mo.conv1d_causal_fwd(h, p[c.name + ".W"], p[c.name + ".b"], s, d=dd)
        acts.append(h)
    assert [a.shape[1] for a in acts] == [T] * 5
    pooled = mo.stats_pool_fwd(h)
    s1 = mo.dense_fwd(pooled, p["fc.W"], p["fc.b"])
    z = mo.dense_fwd(s1, p["out.W"], p["out.b"], relu=False)
    logp = mo.log_softmax(z)
    assert np.abs(m(_dev(x)).cpu().numpy() - logp).max() < 1e-4
    ref_loss = mo.sparse_ce_from_logits(logp, y)
    dz = mo.sparse_ce_from_logits_grad(logp, y)
    g = {}
    dh, g["out.W"], g["out.b"] = mo.dense_bwd(s1, p["out.W"], z, dz, relu=False)
    dh, g["fc.W"], g["fc.b"] = mo.dense_bwd(pooled, p["fc.W"], s1, dh)
    dh = mo.stats_pool_bwd(acts[-1], dh)
    for i in range(len(specs) - 1, -1, -1):
        c = m.convs[i]
        dh, g[c.name + ".W"], g[c.name + ".b"] = mo.conv1d_causal_bwd(acts[i], p[c.name + ".W"], acts[i + 1], dh,
                                                                       specs[i][2], need_dx=(i > 0), d=specs[i][3])
    loss, _ = Trainer(m, use_graph=False).loss_and_grads(_dev(x), _dev(y, np.int32))
    assert abs(float(loss) - ref_loss) <= 1e-4 * abs(ref_loss)
    for name, ref in g.items():
        gotg = m.param(name, grad=True).cpu().numpy()
        assert np.abs(gotg - ref).max() <= 1e-3 * max(1e-12, np.abs(ref).max()), name


def test_dnn_matches_oracle_forward_and_backward():
    """reference dnn.py:13-22: time-distributed Dense x4 -> GlobalAveragePooling1D -> Dense -> log_softmax"""
    from lidbox_amd.models import dnn
    from lidbox_amd.train import Trainer
    rng = np.random.default_rng(50)
    B, T, C, N = 5, 41, 24, 6
    x = rng.standard_normal((B, T, C))
    y = rng.integers(0, N, size=B).astype(np.int32)
    m = dnn.create((T, C), N, seed=2)
    w = m.get_weights()
    assert [w["fc_%d.W" % i].shape for i in (1, 2, 3, 4)] == [(24, 200), (200, 400), (400, 600), (600, 800)]   # Dense kernels
    assert w["output.W"].shape == (800, N) and m.count_params() == 24 * 200 + 200 + 200 * 400 + 400 + 400 * 600 + 600 + 600 * 800 + 800 + 800 * N + N
    rb = np.random.default_rng(51)
    m.set_weights({k: rb.standard_normal(v.shape) * 0.1 for k, v in w.items() if k.endswith(".b")})
    p = _oracle_params(m)
    acts, h = [x], x
    for i in (1, 2, 3, 4):
        h = mo.dense_fwd(h, p["fc_%d.W" % i], p["fc_%d.b" % i])
        acts.append(h)
    pooled = mo.global_avg_pool_fwd(h)
    z = mo.dense_fwd(pooled, p["output.W"], p["output.b"], relu=False)
    logp = mo.log_softmax(z)
    assert np.abs(m(_dev(x)).cpu().numpy() - logp).max() < 1e-4
    ref_loss = mo.sparse_ce_from_logits(logp, y)
    dz = mo.sparse_ce_from_logits_grad(logp, y)
    g = {}
    dh, g["output.W"], g["output.b"] = mo.dense_bwd(pooled, p["output.W"], z, dz, relu=False)
    dh = np.repeat(dh[:, None, :], T, axis=1) / T                     # GlobalAveragePooling1D backward
    for i in (4, 3, 2, 1):
        a_in, a_out = acts[i - 1].reshape(-1, acts[i - 1].shape[-1]), acts[i].reshape(-1, acts[i].shape[-1])
        dx, g["fc_%d.W" % i], g["fc_%d.b" % i] = mo.dense_bwd(a_in, p["fc_%d.W" % i], a_out, dh.reshape(-1, dh.shape[-1]))
        dh = dx.reshape(B, T, -1)
    loss, _ = Trainer(m, use_graph=False).loss_and_grads(_dev(x), _dev(y, np.int32))
    assert abs(float(loss) - ref_loss) <= 1e-4 * abs(ref_loss)
    for name, ref in g.items():
        gotg = m.param(name, grad=True).cpu().numpy()
        assert np.abs(gotg - ref).max() <= 1e-3 * max(1e-12, np.abs(ref).max()), name


@pytest.mark.parametrize("shape", [(1, 1, 8), (3, 7, 20), (2, 198, 40), (4, 50, 12)])
def test_variant_models_valid_output_any_shape(shape):
    """reference tests/test_models.py:30-35 (`_assert_valid_model_output`) for the model modules added under SURVEY 8f:
    output [B, num_outputs], no NaN, training in {False, True}"""
    from lidbox_amd.models import dnn, xvector_extended, xvector_freq_attention
    rng = np.random.default_rng(9)
    x = _dev(rng.uniform(-1e3, 1e3, size=shape))
    for module in (dnn, xvector_extended, xvector_freq_attention):
        for n_out in (1, 100):
            m = module.create(shape[1:], n_out, seed=1)
            assert module.loader is module.create
            for t in (False, True):
                yv = m(x, training=t).cpu().numpy()
                assert yv.shape == (shape[0], n_out) and not np.isnan(yv).any(), (module.__name__, t)
