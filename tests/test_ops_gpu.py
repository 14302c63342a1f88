"""
GPU parity of the individual C-ABI kernels (GEMM family, pooling, losses, C_avg, Adam) against the
oracle / float64 numpy on the same seeded inputs.  fp32 MFMA results are compared with a float64
reference at rel 2e-5 of the result scale (fp32 round-off of a K-long fmaf chain).
"""
import ctypes

import numpy as np
import pytest
import torch

from oracle import model_np as mo

pytestmark = pytest.mark.gpu


def _dev(x, dtype=np.float32):
    return torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype))).cuda()


def _rows(t, bs, rs, batch, rpb, off_floats=0):
    from lidbox_amd import _native as nv
    return nv.Rows(t.data_ptr() + 4 * off_floats, bs, rs, batch, rpb)


def _close(got, ref, rel=2e-5):
    scale = max(1e-30, float(np.abs(ref).max()))
    err = float(np.abs(got - ref).max())
    assert err <= rel * scale, (err, scale)


# asymmetric operands everywhere (a transposed result must not pass)
@pytest.mark.parametrize("M,K,N", [(128, 16, 128), (200, 200, 512), (33, 1536, 512), (1000, 257, 40),
                                   (5, 3, 7), (256, 3000, 512), (130, 64, 1500), (1, 1, 1)])
def test_gemm_nn_plain(M, K, N):
    from lidbox_amd import _native as nv
    rng = np.random.default_rng(M * 7 + K)
    A, B, bias = rng.standard_normal((M, K)), rng.standard_normal((K, N)), rng.standard_normal(N)
    a, b, bi = _dev(A), _dev(B), _dev(bias)
    c = torch.full((M, N), 7.0, device="cuda")
    st = nv.current_stream()
    nv.check(nv.lib.lidbox_gemm_nn(_rows(a, 0, K, 1, M), nv.ptr(b), N, _rows(c, 0, N, 1, M), K, N, nv.EPI_NONE, None, None, 0, st))
    _close(c.cpu().numpy(), A @ B)
    nv.check(nv.lib.lidbox_gemm_nn(_rows(a, 0, K, 1, M), nv.ptr(b), N, _rows(c, 0, N, 1, M), K, N, nv.EPI_BIAS_RELU,
                                   nv.ptr(bi), None, 0, st))
    _close(c.cpu().numpy(), np.maximum(A @ B + bias, 0))
    nv.check(nv.lib.lidbox_gemm_nn(_rows(a, 0, K, 1, M), nv.ptr(b), N, _rows(c, 0, N, 1, M), K, N, nv.EPI_BIAS,
                                   nv.ptr(bi), None, 0, st))
    _close(c.cpu().numpy(), A @ B + bias)
    # with a split-K workspace (small-M problems split along K; epilogue fused into the reduce)
    wsb = max(16, nv.lib.lidbox_gemm_rows_workspace(M, N, K))
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    c.fill_(-3.0)
    nv.check(nv.lib.lidbox_gemm_nn(_rows(a, 0, K, 1, M), nv.ptr(b), N, _rows(c, 0, N, 1, M), K, N, nv.EPI_BIAS_RELU,
                                   nv.ptr(bi), nv.ptr(ws), wsb, st))
    _close(c.cpu().numpy(), np.maximum(A @ B + bias, 0))


@pytest.mark.parametrize("M,K,N", [(128, 512, 1024), (99, 512, 1536), (7, 5, 3), (256, 4, 512), (300, 100, 3000)])
def test_gemm_nt_and_epilogues(M, K, N):
    from lidbox_amd import _native as nv
    rng = np.random.default_rng(M + K * 3)
    A, Bt = rng.standard_normal((M, K)), rng.standard_normal((N, K))
    mask = rng.standard_normal((M, N))
    a, b, mk = _dev(A), _dev(Bt), _dev(mask)
    c = torch.zeros((M, N), device="cuda")
    st = nv.current_stream()
    ref = A @ Bt.T
    nv.check(nv.lib.lidbox_gemm_nt(_rows(a, 0, K, 1, M), nv.ptr(b), K, _rows(c, 0, N, 1, M), K, N, nv.EPI_NONE, None, None, 0, st))
    _close(c.cpu().numpy(), ref)
    nv.check(nv.lib.lidbox_gemm_nt(_rows(a, 0, K, 1, M), nv.ptr(b), K, _rows(c, 0, N, 1, M), K, N, nv.EPI_RELU_MASK,
                                   nv.ptr(mk), None, 0, st))
    _close(c.cpu().numpy(), ref * (mask > 0))
    c.fill_(1.5)
    nv.check(nv.lib.lidbox_gemm_nt(_rows(a, 0, K, 1, M), nv.ptr(b), K, _rows(c, 0, N, 1, M), K, N, nv.EPI_ACCUM, None, None, 0, st))
    _close(c.cpu().numpy(), ref + 1.5)
    c.fill_(-2.0)
    nv.check(nv.lib.lidbox_gemm_nt(_rows(a, 0, K, 1, M), nv.ptr(b), K, _rows(c, 0, N, 1, M), K, N,
                                   nv.EPI_ACCUM_RELU_MASK, nv.ptr(mk), None, 0, st))
    _close(c.cpu().numpy(), ref * (mask > 0) - 2.0)
    wsb = max(16, nv.lib.lidbox_gemm_rows_workspace(M, N, K))
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    c.fill_(-2.0)
    nv.check(nv.lib.lidbox_gemm_nt(_rows(a, 0, K, 1, M), nv.ptr(b), K, _rows(c, 0, N, 1, M), K, N,
                                   nv.EPI_ACCUM_RELU_MASK, nv.ptr(mk), nv.ptr(ws), wsb, st))
    _close(c.cpu().numpy(), ref * (mask > 0) - 2.0)


@pytest.mark.parametrize("plan", ["128,128,1", "128,64,1", "64,128,1", "64,64,1", "128,128,1,8", "128,64,1,8", "64,64,3", "128,64,2,8"])
@pytest.mark.parametrize("M,K,N", [(1500, 200, 512), (777, 520, 200), (130, 36, 68)])
def test_every_rows_decomposition_gives_the_same_product(plan, M, K, N, monkeypatch):
    """every tile shape / split count / 4- and 8-wave kernel the planner (cost model or csrc/gemm_tuned.h) can pick,
    forced through the tuning override, on shapes with ragged edges: nn with bias + ReLU, nt with accumulate + mask"""
    from lidbox_amd import _native as nv
    monkeypatch.setenv("LIDBOX_GEMM_PLAN", plan)
    rng = np.random.default_rng(M + N)
    A, Bm, Bt = rng.standard_normal((M, K)), rng.standard_normal((K, N)), rng.standard_normal((N, K))
    bias, mask = rng.standard_normal(N), rng.standard_normal((M, N))
    a, b, bt, bi, mk = _dev(A), _dev(Bm), _dev(Bt), _dev(bias), _dev(mask)
    st = nv.current_stream()
    ws = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    c = torch.zeros((M, N), device="cuda")
    nv.check(nv.lib.lidbox_gemm_nn(_rows(a, 0, K, 1, M), nv.ptr(b), N, _rows(c, 0, N, 1, M), K, N, nv.EPI_BIAS_RELU, nv.ptr(bi),
                                   nv.ptr(ws), ws.numel(), st))
    _close(c.cpu().numpy(), np.maximum(A @ Bm + bias, 0))
    c.fill_(-2.0)
    nv.check(nv.lib.lidbox_gemm_nt(_rows(a, 0, K, 1, M), nv.ptr(bt), K, _rows(c, 0, N, 1, M), K, N, nv.EPI_ACCUM_RELU_MASK,
                                   nv.ptr(mk), nv.ptr(ws), ws.numel(), st))
    _close(c.cpu().numpy(), (A @ Bt.T) * (mask > 0) - 2.0)
    # batched rows with a gap between utterances (the conv layout): 5 utterances of 30 rows in a [5, 33, N] buffer
    if M >= 150:
        Bn, R, Rp = 5, 30, 33
        cb = torch.full((Bn, Rp, N), 7.0, device="cuda")
        mkb = torch.zeros((Bn, Rp, N), device="cuda")
        mkb[:, 3:, :] = mk[:Bn * R].reshape(Bn, R, N)
        cv, mv = cb[:, 3:, :], mkb[:, 3:, :]
        Cd = nv.Rows(cv.data_ptr(), Rp * N, N, Bn, R)
        nv.check(nv.lib.lidbox_gemm_nt(_rows(a, 0, K, 1, Bn * R), nv.ptr(bt), K, Cd, K, N, nv.EPI_ACCUM_RELU_MASK,
                                       nv.C.c_void_p(mv.data_ptr()), nv.ptr(ws), ws.numel(), st))
        got = cb.cpu().numpy()
        _close(got[:, 3:, :].reshape(Bn * R, N), (A[:Bn * R] @ Bt.T) * (mask[:Bn * R] > 0) + 7.0)
        assert np.all(got[:, :3, :] == 7.0)                     # the rows between utterances are not touched


@pytest.mark.parametrize("M,K1,N", [(4096, 200, 512), (1000, 1536, 512), (256, 3000, 512), (50, 7, 3), (8448, 512, 1500)])
def test_gemm_tn_and_colsum(M, K1, N):
    from lidbox_amd import _native as nv
    rng = np.random.default_rng(M + N)
    A, B = rng.standard_normal((M, K1)), rng.standard_normal((M, N))
    a, b = _dev(A), _dev(B)
    c = torch.full((K1, N), 3.0, device="cuda")
    st = nv.current_stream()
    wsb = nv.lib.lidbox_gemm_tn_workspace(M, K1, N)
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    nv.check(nv.lib.lidbox_gemm_tn(_rows(a, 0, K1, 1, M), _rows(b, 0, N, 1, M), nv.ptr(c), N, K1, N, 0, None, nv.ptr(ws),
                                   wsb, st))
    ref = A.T @ B
    _close(c.cpu().numpy(), ref)
    nv.check(nv.lib.lidbox_gemm_tn(_rows(a, 0, K1, 1, M), _rows(b, 0, N, 1, M), nv.ptr(c), N, K1, N, 1, None, nv.ptr(ws),
                                   wsb, st))
    _close(c.cpu().numpy(), 2 * ref)
    # determinism: bit-identical on a second run
    c2 = torch.empty_like(c)
    nv.check(nv.lib.lidbox_gemm_tn(_rows(a, 0, K1, 1, M), _rows(b, 0, N, 1, M), nv.ptr(c2), N, K1, N, 0, None, nv.ptr(ws),
                                   wsb, st))
    c3 = torch.empty_like(c)
    nv.check(nv.lib.lidbox_gemm_tn(_rows(a, 0, K1, 1, M), _rows(b, 0, N, 1, M), nv.ptr(c3), N, K1, N, 0, None, nv.ptr(ws),
                                   wsb, st))
    assert torch.equal(c2, c3)
    # bias gradient fused into the wgrad kernel (column sums of B from the tiles already in LDS)
    bg = torch.full((N,), 9.0, device="cuda")
    nv.check(nv.lib.lidbox_gemm_tn(_rows(a, 0, K1, 1, M), _rows(b, 0, N, 1, M), nv.ptr(c3), N, K1, N, 0, nv.ptr(bg),
                                   nv.ptr(ws), wsb, st))
    _close(bg.cpu().numpy(), B.sum(axis=0))
    assert torch.equal(c2, c3)
    csb = nv.lib.lidbox_colsum_workspace(M, N)
    cws = torch.empty(csb, dtype=torch.uint8, device="cuda")
    out = torch.zeros(N, device="cuda")
    nv.check(nv.lib.lidbox_colsum(_rows(b, 0, N, 1, M), N, nv.ptr(out), 0, nv.ptr(cws), csb, st))
    _close(out.cpu().numpy(), B.sum(axis=0))


@pytest.mark.parametrize("B,T,C,k,s,Co", [(3, 198, 40, 5, 1, 512), (2, 198, 512, 3, 2, 512), (2, 99, 512, 3, 3, 512),
                                          (4, 37, 12, 7, 2, 500), (2, 5, 13, 3, 2, 9), (3, 1, 7, 5, 1, 11),
                                          (2, 9, 8, 1, 2, 16)])
def test_conv1d_causal_fwd_bwd_via_implicit_rows(B, T, C, k, s, Co):
    """a single causal strided Conv1D layer through the engine's implicit-row GEMMs vs the oracle"""
    from lidbox_amd.models.tdnn import ConvSpec, DenseSpec, SequentialTDNN
    rng = np.random.default_rng(B * 100 + T)
    # two conv layers so that the dgrad of the second is exercised; tiny dense head
    m = SequentialTDNN((T, C), [ConvSpec("c0", C, 1, 1), ConvSpec("c1", Co, k, s)], "stats", [DenseSpec("out", 3, relu=False)],
                       seed=1)
    x = rng.standard_normal((B, T, C))
    W0 = np.eye(C)[None] + 0.0 * rng.standard_normal((1, C, C))
    b0 = np.abs(rng.standard_normal(C)) * 0.0 + 0.5
    W1 = rng.standard_normal((k, C, Co)) * 0.2
    b1 = rng.standard_normal(Co) * 0.1
    Wd = rng.standard_normal((2 * Co, 3)) * 0.1
    bd = rng.standard_normal(3) * 0.1
    m.set_weights({"c0.W": W0, "c0.b": b0, "c1.W": W1, "c1.b": b1, "out.W": Wd, "out.b": bd})
    ws = m.workspace(B, T)
    ws.input_view().copy_(_dev(x))
    logp = m.forward_ws(ws).cpu().numpy()
    # oracle forward
    h0 = mo.conv1d_causal_fwd(x, W0, b0, 1)
    h1 = mo.conv1d_causal_fwd(h0, W1, b1, s)
    pooled = mo.stats_pool_fwd(h1)
    z = pooled @ Wd + bd
    ref = mo.log_softmax(z)
    _close(logp, ref, 5e-5)
    # backward from a random dz
    dz = rng.standard_normal((B, 3))
    ws.dh[-1].copy_(_dev(dz))
    m.backward_ws(ws)
    dpool, dWd, dbd = mo.dense_bwd(pooled, Wd, z, dz, relu=False)
    dh1 = mo.stats_pool_bwd(h1, dpool)
    dh0, dW1, db1 = mo.conv1d_causal_bwd(h0, W1, h1, dh1, s)
    _, dW0, db0 = mo.conv1d_causal_bwd(x, W0, h0, dh0, 1, need_dx=False)
    _close(m.param("out.W", True).cpu().numpy(), dWd, 5e-5)
    _close(m.param("out.b", True).cpu().numpy(), dbd, 5e-5)
    _close(m.param("c1.W", True).cpu().numpy(), dW1, 5e-5)
    _close(m.param("c1.b", True).cpu().numpy(), db1, 5e-5)
    _close(m.param("c0.W", True).cpu().numpy(), dW0, 1e-4)
    _close(m.param("c0.b", True).cpu().numpy(), db0, 1e-4)


def test_stats_and_avg_pool():
    from lidbox_amd import _native as nv
    rng = np.random.default_rng(5)
    for (B, T, C) in [(3, 33, 1500), (2, 1, 5), (4, 7, 64), (1, 100, 3)]:
        x = rng.standard_normal((B, T, C)) * 2 + 0.3
        xd = _dev(x)
        out = torch.zeros((B, 2 * C), device="cuda")
        st = nv.current_stream()
        nv.check(nv.lib.lidbox_stats_pool_fwd(nv.ptr(xd), B, T, C, T * C, C, nv.ptr(out), st))
        ref = mo.stats_pool_fwd(x)
        _close(out.cpu().numpy(), ref, 1e-5)
        dout = rng.standard_normal((B, 2 * C))
        dx = torch.zeros_like(xd)
        nv.check(nv.lib.lidbox_stats_pool_bwd(nv.ptr(xd), nv.ptr(out), nv.ptr(_dev(dout)), B, T, C, T * C, C, 0,
                                              nv.ptr(dx), st))
        _close(dx.cpu().numpy(), mo.stats_pool_bwd(x, dout), 1e-4)
        nv.check(nv.lib.lidbox_stats_pool_bwd(nv.ptr(xd), nv.ptr(out), nv.ptr(_dev(dout)), B, T, C, T * C, C, 1,
                                              nv.ptr(dx), st))
        _close(dx.cpu().numpy(), mo.stats_pool_bwd(x, dout) * (x > 0), 1e-4)
        # the variant that also writes the bf16 shadow of dx (rows padded to 8 channels), with and without the fp32 output
        Cp = (C + 7) // 8 * 8
        for keep32 in (True, False):
            dx2 = torch.full_like(xd, 7.0)
            sh = torch.zeros((B, T, Cp), dtype=torch.bfloat16, device="cuda")
            nv.check(nv.lib.lidbox_stats_pool_bwd_shadow(nv.ptr(xd), nv.ptr(out), nv.ptr(_dev(dout)), B, T, C, T * C, C, 1,
                                                         nv.ptr(dx2) if keep32 else None, nv.ptr(sh), T * Cp, Cp, st))
            assert torch.equal(sh[:, :, :C], dx.bfloat16()) and not sh[:, :, C:].any()
            assert torch.equal(dx2, dx) if keep32 else bool((dx2 == 7.0).all())
        avg = torch.zeros((B, C), device="cuda")
        nv.check(nv.lib.lidbox_avg_pool_fwd(nv.ptr(xd), B, T, C, T * C, C, nv.ptr(avg), st))
        _close(avg.cpu().numpy(), x.mean(axis=1), 1e-5)
    # T = 1: std = sqrt(1e-10) = 1e-5 (xvector.py:22,34)
    x = _dev(rng.standard_normal((2, 1, 5)))
    out = torch.zeros((2, 10), device="cuda")
    nv.check(nv.lib.lidbox_stats_pool_fwd(nv.ptr(x), 2, 1, 5, 5, 5, nv.ptr(out), nv.current_stream()))
    assert np.allclose(out[:, 5:].cpu().numpy(), 1e-5, rtol=1e-6)


@pytest.mark.parametrize("B,T,C,Cp", [(3, 33, 1500, 1504), (2, 1, 8, 8), (5, 40, 64, 64), (4, 12, 260, 264), (3, 20, 36, 36), (2, 33, 1500, 1500),
                                      (7, 36, 520, 528)])
def test_stats_pool_over_a_bf16_shadow(B, T, C, Cp):
    """lidbox_stats_pool_fwd_bf16 / _bwd_bf16 (the bf16 policy's all-shadow mode: the pooling reads the last frame layer's bfloat16
    shadow, row pitch Cp >= C a multiple of 4 channels): bit-identical to the fp32 kernels fed the shadow's values, hence within the fp32
    kernels' tolerance of the oracle on those values; shapes the register kernels do not cover are refused"""
    from lidbox_amd import _native as nv
    rng = np.random.default_rng(B * 100 + T)
    x16 = torch.zeros((B, T, Cp), dtype=torch.bfloat16, device="cuda")
    x16[:, :, :C] = _dev(rng.standard_normal((B, T, C)) * 2 + 0.3).bfloat16()
    x32 = x16[:, :, :C].float().contiguous()
    st = nv.current_stream()
    out, ref = torch.zeros((B, 2 * C), device="cuda"), torch.zeros((B, 2 * C), device="cuda")
    nv.check(nv.lib.lidbox_stats_pool_fwd_bf16(nv.ptr(x16), B, T, C, T * Cp, Cp, nv.ptr(out), st))
    nv.check(nv.lib.lidbox_stats_pool_fwd(nv.ptr(x32), B, T, C, T * C, C, nv.ptr(ref), st))
    assert torch.equal(out, ref)
    _close(out.cpu().numpy(), mo.stats_pool_fwd(x32.cpu().double().numpy()), 1e-5)
    dout = _dev(rng.standard_normal((B, 2 * C)))
    for mask in (0, 1):
        sh = torch.full((B, T, Cp), 3.0, dtype=torch.bfloat16, device="cuda")
        sh_ref = torch.full((B, T, Cp), 3.0, dtype=torch.bfloat16, device="cuda")
        nv.check(nv.lib.lidbox_stats_pool_bwd_bf16(nv.ptr(x16), nv.ptr(out), nv.ptr(dout), B, T, C, T * Cp, Cp, mask, nv.ptr(sh), T * Cp, Cp, st))
        nv.check(nv.lib.lidbox_stats_pool_bwd_shadow(nv.ptr(x32), nv.ptr(out), nv.ptr(dout), B, T, C, T * C, C, mask, None, nv.ptr(sh_ref),
                                                     T * Cp, Cp, st))
        assert torch.equal(sh, sh_ref)                              # pad columns untouched by both
    with pytest.raises(ValueError):                                 # 41 frames: not a register-kernel shape
        nv.check(nv.lib.lidbox_stats_pool_fwd_bf16(nv.ptr(x16), B, 41, C, T * Cp, Cp, nv.ptr(out), st))
    with pytest.raises(ValueError):                                 # odd row stride
        nv.check(nv.lib.lidbox_stats_pool_fwd_bf16(nv.ptr(x16), B, T, C, T * Cp, Cp + 2, nv.ptr(out), st))


def test_log_softmax_nll():
    from lidbox_amd import _native as nv
    rng = np.random.default_rng(6)
    for (B, N) in [(256, 4), (37, 100), (1, 1), (5, 130)]:
        z = rng.standard_normal((B, N)) * 3
        y = rng.integers(0, N, size=B).astype(np.int32)
        zd = _dev(z)
        logp = torch.zeros_like(zd)
        st = nv.current_stream()
        nv.check(nv.lib.lidbox_log_softmax_fwd(nv.ptr(zd), B, N, nv.ptr(logp), st))
        ref = mo.log_softmax(z)
        assert np.abs(logp.cpu().numpy() - ref).max() < 1e-5
        loss = torch.zeros(4, device="cuda")
        dz = torch.zeros_like(zd)
        nv.check(nv.lib.lidbox_nll_fwd_bwd(nv.ptr(logp), nv.ptr(_dev(y, np.int32)), B, N, 1.0 / B, nv.ptr(loss),
                                           nv.ptr(dz), st))
        assert abs(float(loss[0]) - mo.sparse_ce_from_logits(ref, y)) < 1e-5
        assert np.abs(dz.cpu().numpy() - mo.sparse_ce_from_logits_grad(ref, y)).max() < 1e-6


def test_ap_loss_known_answers_and_grad():
    from lidbox_amd.losses import SparseAngularProximity
    N, D = 3, 100
    y_true = np.array([0, 1, 1, 1, 0, 2, 1, 2], np.int32)
    cases = {(0, 1, 1, 1, 0, 2, 1, 2): 0.3442058, (0, 1, 1, 2, 0, 2, 1, 2): 0.4671672,
             (1, 2, 0, 2, 1, 1, 0, 1): 1.3278971}                       # reference losses.py:61-97 demo
    ap = SparseAngularProximity(N, D)
    for pred, expect in cases.items():
        z = _dev(np.eye(D)[list(pred)])
        got = float(ap(_dev(y_true, np.int32), z))
        assert abs(got - expect) < 2e-6, (got, expect)
    rng = np.random.default_rng(13)
    z = mo.l2_normalize(rng.standard_normal((64, 40)))
    y = rng.integers(0, 17, size=64).astype(np.int32)
    ap = SparseAngularProximity(17, 40, delta_weight=1.7)
    loss, dz = ap.loss_and_grad(_dev(y, np.int32), _dev(z))
    assert abs(float(loss) - mo.ap_loss(y, z, 17, 1.7)) < 1e-5
    _close(dz.cpu().numpy(), mo.ap_loss_grad(y, z, 17, 1.7), 1e-4)
    assert np.abs(ap.call(_dev(y, np.int32), _dev(z)).cpu().numpy() - mo.ap_loss_per_example(y, z, 17, 1.7)).max() < 1e-5
    # theta / predict (losses.py:42-52) on the library's kernels: acos of the first N coordinates, and its negation
    th, pr = ap.theta(_dev(z)).cpu().numpy(), ap.predict(_dev(z)).cpu().numpy()
    assert th.shape == pr.shape == (64, 17)
    assert np.abs(th - np.arccos(z[:, :17])).max() < 2e-6 and np.array_equal(pr, -th)
    with pytest.raises(ValueError):
        SparseAngularProximity(5, 4)
    with pytest.raises(ValueError):
        SparseAngularProximity(5, 8, delta_weight=0.0)


def test_ap_head_one_launch_equals_the_four_separate_calls():
    """lidbox_ap_head_fwd_bwd (what the train step runs for the angular-proximity loss): normalised rows, per-example loss,
    gradient through the normalisation and predict() scores, bit for bit what lidbox_l2_normalize_fwd -> lidbox_ap_loss_fwd_bwd
    -> lidbox_l2_normalize_bwd -> lidbox_neg_acos give; against the float64 oracle as well; an out-of-range label gives a NaN
    loss and a zero gradient row in both forms"""
    from lidbox_amd import _native as nv
    rng = np.random.default_rng(21)
    st = nv.current_stream()
    for B, D, N, delta in ((37, 512, 100, 1.0), (5, 40, 17, 1.7), (64, 100, 100, 0.5), (3, 4096, 3, 2.0)):
        x = rng.standard_normal((B, D)).astype(np.float32)
        y = rng.integers(0, N, size=B).astype(np.int32)
        if B > 4:
            y[2] = N + 3                                               # invalid label
        xd, yd = _dev(x), _dev(y, np.int32)
        zn, dzn, dx = torch.zeros_like(xd), torch.zeros_like(xd), torch.zeros_like(xd)
        per, sc = torch.zeros(B, device="cuda"), torch.zeros((B, N), device="cuda")
        nv.check(nv.lib.lidbox_l2_normalize_fwd(nv.ptr(xd), B, D, nv.ptr(zn), st))
        nv.check(nv.lib.lidbox_ap_loss_fwd_bwd(nv.ptr(zn), nv.ptr(yd), B, D, N, delta, 1.0 / B, nv.ptr(per), nv.ptr(dzn), st))
        nv.check(nv.lib.lidbox_l2_normalize_bwd(nv.ptr(xd), nv.ptr(dzn), B, D, nv.ptr(dx), st))
        nv.check(nv.lib.lidbox_neg_acos(nv.ptr(zn), B, D, N, nv.ptr(sc), st))
        zn2, dx2 = torch.full_like(xd, 7.0), torch.full_like(xd, 7.0)
        per2, sc2 = torch.full((B,), 7.0, device="cuda"), torch.full((B, N), 7.0, device="cuda")
        nv.check(nv.lib.lidbox_ap_head_fwd_bwd(nv.ptr(xd), nv.ptr(yd), B, D, N, delta, 1.0 / B, nv.ptr(zn2), nv.ptr(per2), nv.ptr(dx2), nv.ptr(sc2), st))
        assert torch.equal(zn2, zn) and torch.equal(dx2, dx) and torch.equal(sc2, sc)
        assert torch.equal(torch.nan_to_num(per2, nan=-1.0), torch.nan_to_num(per, nan=-1.0))
        ok = y < N
        assert bool(torch.isnan(per2[torch.from_numpy(~ok).cuda()]).all()) and float(dx2[torch.from_numpy(~ok).cuda()].abs().sum()) == 0.0
        # float64 oracle on the valid rows: loss per example and the gradient with respect to the un-normalised rows (autograd)
        xt = torch.tensor(x[ok].astype(np.float64), requires_grad=True)
        from oracle import torch_ref as tref
        z64 = torch.nn.functional.normalize(xt, dim=1)
        assert np.abs(per2.cpu().numpy()[ok] - mo.ap_loss_per_example(y[ok], z64.detach().numpy(), N, delta)).max() < 1e-4
        (tref.ap_loss(torch.from_numpy(y[ok].astype(np.int64)), z64, N, delta) * ok.sum() / B).backward()
        _close(dx2.cpu().numpy()[ok], xt.grad.numpy(), 1e-4)
        # optional outputs may be NULL
        nv.check(nv.lib.lidbox_ap_head_fwd_bwd(nv.ptr(xd), nv.ptr(yd), B, D, N, delta, 1.0 / B, None, nv.ptr(per2), None, None, st))
    assert nv.lib.lidbox_ap_head_fwd_bwd(nv.ptr(xd), nv.ptr(yd), 3, 8192, 3, 1.0, 1.0, None, nv.ptr(per2), None, None, st) == -1   # D > 4096


def test_l2_normalize():
    from lidbox_amd import _native as nv
    rng = np.random.default_rng(14)
    x = rng.standard_normal((33, 512))
    g = rng.standard_normal((33, 512))
    xd, gd = _dev(x), _dev(g)
    y, dx = torch.zeros_like(xd), torch.zeros_like(xd)
    st = nv.current_stream()
    nv.check(nv.lib.lidbox_l2_normalize_fwd(nv.ptr(xd), 33, 512, nv.ptr(y), st))
    _close(y.cpu().numpy(), mo.l2_normalize(x), 1e-5)
    nv.check(nv.lib.lidbox_l2_normalize_bwd(nv.ptr(xd), nv.ptr(gd), 33, 512, nv.ptr(dx), st))
    xt = torch.tensor(x, requires_grad=True)
    (torch.nn.functional.normalize(xt, dim=1) * torch.tensor(g)).sum().backward()
    _close(dx.cpu().numpy(), xt.grad.numpy(), 1e-4)


def test_cavg_demo_and_random():
    from lidbox_amd.metrics import AverageDetectionCost, SparseAverageDetectionCost
    true_pos = np.array([[1, 0, 0], [0, 1, 0], [0, 1, 0], [0, 1, 0], [1, 0, 0], [0, 0, 1], [0, 1, 0], [0, 0, 1]], np.float32)
    with np.errstate(divide="ignore"):
        pred = np.log(np.array([[.1, .2, .9], [.9, .2, .0], [.1, .9, .0], [.2, .8, .5], [.6, .3, .1], [.1, .0, .7],
                                [.1, .0, .7], [.9, .1, .0]], np.float32))
    th = np.log(np.array([0.05, 0.4, 0.6, 0.95], np.float32))
    c = AverageDetectionCost(3, th)
    c.update_state(_dev(true_pos), _dev(pred))
    res, per = c.result(return_per_threshold=True)
    assert abs(float(res) - 0.375) < 1e-6                                   # reference metrics.py demo
    assert np.allclose(per.cpu().numpy(), [0.5416667, 0.3958333, 0.375, 0.5], atol=1e-6)
    c.reset_states()
    assert float(c.result()) == 0.0                                         # metrics.py:163-164
    # random, multi-batch streaming, N=100, Th=100 (util.py:76-80 uses 100 thresholds)
    rng = np.random.default_rng(15)
    N, Th = 100, 100
    thr = np.linspace(-5, 0, Th).astype(np.float32)
    g = SparseAverageDetectionCost(N, thr)
    o = mo.SparseAverageDetectionCost(N, thr)
    for _ in range(3):
        s = mo.log_softmax(rng.standard_normal((300, N)) * 2).astype(np.float32)
        y = rng.integers(0, N, size=300)
        g.update_state(_dev(y, np.int64), _dev(s))
        o.update_state(y, s)
    assert torch.equal(g.tp.cpu(), torch.from_numpy(o.tp)) and torch.equal(g.fn.cpu(), torch.from_numpy(o.fn))
    assert torch.equal(g.fp_pairs.cpu(), torch.from_numpy(o.fp_pairs))
    assert torch.equal(g.tn_pairs.cpu(), torch.from_numpy(o.tn_pairs))      # counters are exact integers
    assert abs(float(g.result()) - o.result()) < 1e-6
    with pytest.raises(ValueError):
        AverageDetectionCost(1, th)


@pytest.mark.parametrize("B,N,Th", [(1, 2, 1), (17, 3, 5), (1100, 130, 65), (2048, 14, 100), (33, 257, 130), (512, 50, 100)])
@pytest.mark.parametrize("nz,unit", [(None, None), ("3", "16"), ("1", "64"), ("7", "32")])
def test_cavg_counters_exact_at_edge_shapes(B, N, Th, nz, unit, monkeypatch):
    """lidbox_cavg_update (sixteen waves, one shared LDS copy, label chunks along grid.z): the four counter arrays are the oracle's
    integers exactly -- one example, more classes than one LDS chunk holds (> 128), threshold counts off the 64-lane grid, batches past
    one pass of 16 waves x 64 examples, two updates on top of each other, and every forced chunk count / walk unit"""
    from lidbox_amd.metrics import SparseAverageDetectionCost
    for k, v in (("LIDBOX_CAVG_NZ", nz), ("LIDBOX_CAVG_UNIT", unit)):
        if v is not None:
            monkeypatch.setenv(k, v)
    rng = np.random.default_rng(B * 7 + N)
    thr = np.sort(rng.uniform(-5, 0, Th)).astype(np.float32)
    g = SparseAverageDetectionCost(N, thr)
    o = mo.SparseAverageDetectionCost(N, thr)
    for _ in range(2):
        s = mo.log_softmax(rng.standard_normal((B, N)) * 2).astype(np.float32)
        s[rng.integers(0, B), rng.integers(0, N)] = thr[rng.integers(0, Th)]          # a score exactly on a threshold (>= counts)
        y = rng.integers(0, N, size=B)
        g.update_state(_dev(y, np.int64), _dev(s))
        o.update_state(y, s)
    for name in ("tp", "fn", "fp_pairs", "tn_pairs"):
        assert torch.equal(getattr(g, name).cpu(), torch.from_numpy(getattr(o, name))), name
    assert abs(float(g.result()) - o.result()) < 1e-6


def test_adam_matches_keras_formula():
    from lidbox_amd import _native as nv
    rng = np.random.default_rng(16)
    n = 1003
    p = {"w": rng.standard_normal(n)}
    m, v = {"w": np.zeros(n)}, {"w": np.zeros(n)}
    pd = _dev(np.concatenate([p["w"], np.zeros(1)]))[:n]
    md, vd = torch.zeros(n + 1, device="cuda")[:n], torch.zeros(n + 1, device="cuda")[:n]
    state = torch.zeros(16, dtype=torch.uint8, device="cuda")
    for t in range(1, 6):
        g = rng.standard_normal(n)
        gd = _dev(np.concatenate([g, np.zeros(1)]))[:n]
        mo.adam_step(p, {"w": g * 0.5}, m, v, t)
        nv.check(nv.lib.lidbox_adam_step(nv.ptr(pd), nv.ptr(gd), nv.ptr(md), nv.ptr(vd), n, 1e-3, 0.9, 0.999, 1e-7, 0.5,
                                         nv.ptr(state), nv.current_stream()))
        assert np.abs(pd.cpu().numpy() - p["w"]).max() < 2e-6
    assert int(state[:8].view(torch.int64).item()) == 5


def test_abi_rejects_bad_arguments():
    from lidbox_amd import _native as nv
    x = torch.zeros(16, device="cuda")
    assert nv.lib.lidbox_gemm_nn(nv.Rows(None, 0, 4, 1, 4), nv.ptr(x), 4, nv.Rows(x.data_ptr(), 0, 4, 1, 4), 4, 4, 0,
                                 None, None, 0, None) == -1
    assert "lidbox_gemm_nn" in nv.last_error()
    assert nv.lib.lidbox_stats_pool_fwd(None, 1, 1, 1, 1, 1, None, None) == -1
    assert nv.lib.lidbox_ap_loss_fwd_bwd(nv.ptr(x), nv.ptr(x), 1, 2, 3, 1.0, 1.0, nv.ptr(x), None, None) == -1


def test_out_of_range_labels_give_nan_loss_and_zero_gradient_rows():
    """a label outside [0, N) is never used as an index (TF: InvalidArgument on CPU, NaN on GPU)"""
    from lidbox_amd import _native as nv
    rng = np.random.default_rng(60)
    B, N, D = 9, 5, 16
    st = nv.current_stream()
    y = rng.integers(0, N, size=B).astype(np.int32)
    y[2], y[7] = N, -1
    logp = _dev(mo.log_softmax(rng.standard_normal((B, N))))
    loss = torch.zeros(4, device="cuda")
    dz = torch.full((B, N), 5.0, device="cuda")
    nv.check(nv.lib.lidbox_nll_fwd_bwd(nv.ptr(logp), nv.ptr(_dev(y, np.int32)), B, N, 1.0 / B, nv.ptr(loss), nv.ptr(dz), st))
    assert np.isnan(float(loss[0]))
    g = dz.cpu().numpy()
    assert not g[2].any() and not g[7].any() and np.isfinite(g).all() and g[0].any()
    z = rng.standard_normal((B, D))
    zn = _dev(z / np.linalg.norm(z, axis=1, keepdims=True))
    per = torch.zeros(B, device="cuda")
    dzn = torch.full((B, D), 5.0, device="cuda")
    nv.check(nv.lib.lidbox_ap_loss_fwd_bwd(nv.ptr(zn), nv.ptr(_dev(y, np.int32)), B, D, N, 16.0, 1.0 / B, nv.ptr(per), nv.ptr(dzn), st))
    p, g = per.cpu().numpy(), dzn.cpu().numpy()
    assert np.isnan(p[2]) and np.isnan(p[7]) and np.isfinite(np.delete(p, [2, 7])).all()
    assert not g[2].any() and not g[7].any() and np.isfinite(g).all()


def test_mean_neg_acos_copy_zero_2d():
    from lidbox_amd import _native as nv
    rng = np.random.default_rng(61)
    st = nv.current_stream()
    for n in (1, 255, 4096, 10001):
        x = rng.standard_normal(n)
        out = torch.zeros(1, device="cuda")
        nv.check(nv.lib.lidbox_mean(nv.ptr(_dev(x)), n, nv.ptr(out), st))
        assert abs(float(out) - x.astype(np.float32).astype(np.float64).mean()) < 1e-5
    z = rng.uniform(-1, 1, size=(37, 20))
    out = torch.zeros((37, 7), device="cuda")
    nv.check(nv.lib.lidbox_neg_acos(nv.ptr(_dev(z)), 37, 20, 7, nv.ptr(out), st))
    assert np.abs(out.cpu().numpy() + np.arccos(z[:, :7].astype(np.float32))).max() < 1e-5
    # pitched copy into / zero fill of a padded layer input [B, pad + T, C]
    B, T, C, pad = 5, 11, 6, 3
    src = _dev(rng.standard_normal((B, T, C)))
    dst = torch.full((B, pad + T, C), 9.0, device="cuda")
    nv.check(nv.lib.lidbox_copy_2d(ctypes.c_void_p(dst.data_ptr() + 4 * pad * C), 4 * (pad + T) * C, nv.ptr(src), 4 * T * C,
                                   4 * T * C, B, st))
    assert torch.equal(dst[:, pad:], src) and bool((dst[:, :pad] == 9.0).all())
    nv.check(nv.lib.lidbox_zero_2d(ctypes.c_void_p(dst.data_ptr() + 4 * (pad + T - 2) * C), 4 * (pad + T) * C, 4 * 2 * C, B, st))
    assert bool((dst[:, -2:] == 0).all()) and torch.equal(dst[:, pad:-2], src[:, :-2])


def test_cmvn_strided_in_place_equals_dense():
    from lidbox_amd import _native as nv
    rng = np.random.default_rng(62)
    st = nv.current_stream()
    B, T, C, pad = 7, 98, 12, 5
    x = rng.standard_normal((B, T, C)) * 3 + 1
    dense_in, dense_out = _dev(x), torch.zeros((B, T, C), device="cuda")
    for nv_flag in (0, 1):
        nv.check(nv.lib.lidbox_cmvn_fwd(nv.ptr(dense_in), B, T, C, nv_flag, nv.ptr(dense_out), st))
        padded = torch.zeros((B, pad + T, C), device="cuda")
        padded[:, pad:] = dense_in
        p = ctypes.c_void_p(padded.data_ptr() + 4 * pad * C)
        nv.check(nv.lib.lidbox_cmvn_strided_fwd(p, B, T, C, (pad + T) * C, nv_flag, p, (pad + T) * C, st))
        assert torch.equal(padded[:, pad:], dense_out) and not bool(padded[:, :pad].any())


def test_spatial_dropout_kernel():
    """whole channels of an utterance dropped for all frames, kept ones scaled by 1/(1-rate); the mask follows the device
    step counter; rate 0 is the identity"""
    from lidbox_amd import _native as nv
    st = nv.current_stream()
    B, T, C, pad, rate = 64, 20, 40, 4, 0.25
    step = torch.zeros(2, dtype=torch.int64, device="cuda")
    masks = []
    for s_ in range(3):
        step[0] = s_
        x = torch.ones((B, pad + T, C), device="cuda")
        m = torch.zeros((B, C), device="cuda")
        p = ctypes.c_void_p(x.data_ptr() + 4 * pad * C)
        nv.check(nv.lib.lidbox_spatial_dropout(p, B, T, C, (pad + T) * C, rate, 1234, nv.ptr(step), nv.ptr(m), st))
        xv = x[:, pad:]
        assert bool((x[:, :pad] == 1).all())                               # rows before the view untouched
        assert bool((xv == xv[:, :1]).all())                               # the same factor for every frame
        assert torch.equal(xv[:, 0], m)
        vals = set(np.unique(m.cpu().numpy()).tolist())
        assert vals == {0.0, np.float32(1.0 / (1.0 - rate)).item()}
        masks.append(m.cpu().numpy())
    keep = np.mean([float((m > 0).mean()) for m in masks])
    assert abs(keep - (1 - rate)) < 0.03                                    # 7680 draws
    assert not np.array_equal(masks[0], masks[1]) and not np.array_equal(masks[1], masks[2])
    x = torch.ones((B, T, C), device="cuda")
    nv.check(nv.lib.lidbox_spatial_dropout(nv.ptr(x), B, T, C, T * C, 0.0, 1, None, None, st))
    assert bool((x == 1).all())
    with pytest.raises(ValueError):
        nv.check(nv.lib.lidbox_spatial_dropout(nv.ptr(x), B, T, C, T * C, 1.0, 1, None, None, st))


@pytest.mark.parametrize("plan", ["128,128,1", "128,64,5", "128,128,3", "64,64,7"])
@pytest.mark.parametrize("M,K1,N", [(3000, 300, 200), (700, 1536, 512), (33, 257, 129)])
def test_every_wgrad_decomposition_gives_the_same_product(plan, M, K1, N, monkeypatch):
    """lidbox_gemm_tn under forced tile shapes / row splits (LIDBOX_GEMM_TN_PLAN)
    on ragged shapes with implicit (gapped) utterance rows; weight gradient and fused bias gradient"""
    from lidbox_amd import _native as nv
    rng = np.random.default_rng(M + K1 + N)
    Bn = 3 if M % 3 == 0 else 1
    rpb, pad = M // Bn, 2
    A = rng.standard_normal((Bn, pad + rpb, K1))            # rows behind two pad rows per utterance
    Bm = rng.standard_normal((Bn, rpb, N))
    a, b = _dev(A), _dev(Bm)
    st = nv.current_stream()
    monkeypatch.setenv("LIDBOX_GEMM_TN_PLAN", plan)
    wsb = nv.lib.lidbox_gemm_tn_workspace(M, K1, N)
    ws = torch.empty(max(16, wsb), dtype=torch.uint8, device="cuda")
    c = torch.full((K1, N), 5.0, device="cuda")
    bg = torch.full((N,), 5.0, device="cuda")
    ra = _rows(a, (pad + rpb) * K1, K1, Bn, rpb, off_floats=pad * K1)
    rb = _rows(b, rpb * N, N, Bn, rpb)
    nv.check(nv.lib.lidbox_gemm_tn(ra, rb, nv.ptr(c), N, K1, N, 0, nv.ptr(bg), nv.ptr(ws), ws.numel(), st))
    A2, B2 = A[:, pad:].reshape(M, K1), Bm.reshape(M, N)
    _close(c.cpu().numpy(), A2.T @ B2)
    _close(bg.cpu().numpy(), B2.sum(axis=0))


@pytest.mark.parametrize("B,K,N,relu", [(256, 512, 4, 1), (37, 512, 4, 1), (16, 64, 3, 0), (1, 8, 1, 1), (300, 512, 32, 1),
                                          (50, 1500, 17, 0), (2048, 512, 4, 1)])
def test_softmax_head_fused_matches_the_separate_ops(B, K, N, relu):
    """lidbox_softmax_head_fwd_bwd = Dense(N) + log_softmax + sparse cross-entropy and their backward, against the float64
    restatement; invalid labels, row counts that are not a multiple of 16, determinism"""
    from lidbox_amd import _native as nv
    rng = np.random.default_rng(B + K + N)
    h = np.maximum(rng.standard_normal((B, K)), 0) if relu else rng.standard_normal((B, K))
    W, b = rng.standard_normal((K, N)) * 0.1, rng.standard_normal(N)
    y = rng.integers(0, N, size=B).astype(np.int32)
    scale = 1.0 / (B + 3)                                        # a global batch larger than the shard
    hd, Wd, bd, yd = _dev(h), _dev(W), _dev(b), _dev(y, np.int32)
    st = nv.current_stream()
    assert nv.lib.lidbox_softmax_head_supported(K, N) == 1
    wsb = nv.lib.lidbox_softmax_head_workspace(B, K, N)
    ws = torch.full((wsb,), 0xff, dtype=torch.uint8, device="cuda")          # the workspace needs no initialisation
    outs = []
    for rep in range(2):
        logp = torch.full((B, N), 9.0, device="cuda")
        loss = torch.zeros(4, device="cuda")
        dW, db, dh = torch.full((K, N), 9.0, device="cuda"), torch.full((N,), 9.0, device="cuda"), torch.full((B, K), 9.0, device="cuda")
        nv.check(nv.lib.lidbox_softmax_head_fwd_bwd(nv.ptr(hd), nv.ptr(Wd), nv.ptr(bd), nv.ptr(yd), B, K, N, scale, relu, nv.ptr(logp),
                                                    nv.ptr(loss), nv.ptr(dW), nv.ptr(db), nv.ptr(dh), nv.ptr(ws), wsb, st))
        outs.append((logp, loss.clone(), dW, db, dh))
    assert all(torch.equal(a, b_) for a, b_ in zip(outs[0], outs[1]))               # deterministic
    logp, loss, dW, db, dh = outs[0]
    h32, W32, b32 = h.astype(np.float32).astype(np.float64), W.astype(np.float32).astype(np.float64), b.astype(np.float32).astype(np.float64)
    ref_logp = mo.log_softmax(h32 @ W32 + b32)
    assert np.abs(logp.cpu().numpy() - ref_logp).max() < 2e-5
    assert abs(float(loss[0]) - mo.sparse_ce_from_logits(ref_logp, y)) < 2e-5
    dz = mo.sparse_ce_from_logits_grad(ref_logp, y) * B * scale                     # the oracle's gradient is of the shard mean
    _close(dW.cpu().numpy(), h32.T @ dz, 2e-5)
    _close(db.cpu().numpy(), dz.sum(axis=0), 2e-5)
    ref_dh = dz @ W32.T
    if relu:
        ref_dh = ref_dh * (h32 > 0)
    _close(dh.cpu().numpy(), ref_dh, 2e-5)
    # a label outside [0, N): NaN loss, zero gradient row, everything else untouched by it
    y2 = y.copy(); y2[0] = N
    nv.check(nv.lib.lidbox_softmax_head_fwd_bwd(nv.ptr(hd), nv.ptr(Wd), nv.ptr(bd), nv.ptr(_dev(y2, np.int32)), B, K, N, scale, relu,
                                                nv.ptr(logp), nv.ptr(loss), nv.ptr(dW), nv.ptr(db), nv.ptr(dh), nv.ptr(ws), wsb, st))
    assert np.isnan(float(loss[0])) and not dh[0].any() and torch.isfinite(dW).all()
    if B > 1:
        _close(dh[1:].cpu().numpy(), ref_dh[1:], 2e-5)


def test_softmax_head_limits():
    from lidbox_amd import _native as nv
    assert nv.lib.lidbox_softmax_head_supported(512, 33) == 0 and nv.lib.lidbox_softmax_head_supported(3000, 32) == 1
    x = torch.zeros(64, device="cuda")
    with pytest.raises(ValueError):
        nv.check(nv.lib.lidbox_softmax_head_fwd_bwd(nv.ptr(x), nv.ptr(x), nv.ptr(x), nv.ptr(x), 1, 8, 33, 1.0, 0, nv.ptr(x), nv.ptr(x),
                                                    nv.ptr(x), nv.ptr(x), None, nv.ptr(x), 256, nv.current_stream()))
    with pytest.raises(ValueError):                              # workspace too small
        nv.check(nv.lib.lidbox_softmax_head_fwd_bwd(nv.ptr(x), nv.ptr(x), nv.ptr(x), nv.ptr(x), 4, 8, 2, 1.0, 0, nv.ptr(x), nv.ptr(x),
                                                    nv.ptr(x), nv.ptr(x), None, nv.ptr(x), 16, nv.current_stream()))


def test_sgd_and_rmsprop_updates():
    """lidbox_sgd_step / lidbox_rmsprop_step on random vectors against the oracle's restatement of the TensorFlow 2.3 dense updates
    (oracle/model_np.py, reference call site keras_utils.py:137-140), five steps each, every variant; the device step counter
    advances and a scheduled rate (lr_now) replaces the constructor's"""
    from lidbox_amd import _native as nv
    rng = np.random.default_rng(31)
    n = 1003
    st = nv.current_stream()
    w0 = rng.standard_normal(n).astype(np.float32)
    grads = [rng.standard_normal(n).astype(np.float32) * 0.1 for _ in range(5)]
    for kw in (dict(momentum=0.0, nesterov=False), dict(momentum=0.9, nesterov=False), dict(momentum=0.6, nesterov=True)):
        p, vel = {"w": w0.astype(np.float64)}, {"w": np.zeros(n)}
        wd, vd, state = _dev(w0), torch.zeros(n, device="cuda"), torch.zeros(16, dtype=torch.uint8, device="cuda")
        for g in grads:
            mo.sgd_step(p, {"w": g.astype(np.float64)}, vel, lr=0.03, **kw)
            nv.check(nv.lib.lidbox_sgd_step(nv.ptr(wd), nv.ptr(_dev(g)), nv.ptr(vd), n, 0.03, kw["momentum"], int(kw["nesterov"]), 1.0, nv.ptr(state), st))
        assert np.abs(wd.cpu().numpy() - p["w"]).max() <= 2e-6, kw
        assert int(state[:8].view(torch.int64).item()) == 5
    for kw in (dict(), dict(momentum=0.7), dict(centered=True), dict(centered=True, momentum=0.3, rho=0.8, eps=1e-5)):
        p, rms, mg, mm = {"w": w0.astype(np.float64)}, {"w": np.zeros(n)}, {"w": np.zeros(n)}, {"w": np.zeros(n)}
        wd, state = _dev(w0), torch.zeros(16, dtype=torch.uint8, device="cuda")
        rd, gd, md = (torch.zeros(n, device="cuda") for _ in range(3))
        for g in grads:
            mo.rmsprop_step(p, {"w": g.astype(np.float64)}, rms, mg, mm, lr=2e-3, **kw)
            nv.check(nv.lib.lidbox_rmsprop_step(nv.ptr(wd), nv.ptr(_dev(g)), nv.ptr(rd), nv.ptr(gd), nv.ptr(md), n, 2e-3, kw.get("rho", 0.9),
                                                kw.get("momentum", 0.0), kw.get("eps", 1e-7), int(kw.get("centered", False)), 1.0, nv.ptr(state), st))
        assert np.abs(wd.cpu().numpy() - p["w"]).max() <= 5e-6, kw
    # a scheduled rate in the state's lr_now replaces lr (what Trainer's lr_schedule writes)
    wd, state = _dev(w0), torch.zeros(16, dtype=torch.uint8, device="cuda")
    state[12:16].view(torch.float32).fill_(0.5)
    nv.check(nv.lib.lidbox_sgd_step(nv.ptr(wd), nv.ptr(_dev(grads[0])), None, n, 0.03, 0.0, 0, 1.0, nv.ptr(state), st))
    assert np.abs(wd.cpu().numpy() - (w0 - 0.5 * grads[0])).max() <= 1e-6
    assert nv.lib.lidbox_sgd_step(nv.ptr(wd), nv.ptr(wd), None, n, 0.03, 0.9, 0, 1.0, nv.ptr(state), st) == -1        # momentum without a buffer


def test_adam_prepare_job_plus_apply_equals_adam_step():
    """lidbox_adam_prepare_job run by lidbox_reduce_jobs_run (together with a wgrad-style slice sum in the same launch) +
    lidbox_adam_apply == lidbox_adam_step bit for bit over three steps, with and without a scheduled rate; GEMM carriers
    refuse the job kind.  Reference: tf.keras.optimizers.Adam under fit, lidbox/models/keras_utils.py:135-140,198-203."""
    from lidbox_amd import _native as nv
    import ctypes
    rng = np.random.default_rng(4)
    n = 4099
    p0, g = rng.standard_normal(n).astype(np.float32), rng.standard_normal(n).astype(np.float32)
    st = nv.current_stream()
    res = []
    for split in (False, True):
        pd, gd = torch.from_numpy(p0.copy()).cuda(), torch.from_numpy(g).cuda()
        md, vd = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
        state = torch.zeros(16, dtype=torch.uint8, device="cuda")
        P = torch.from_numpy(rng.standard_normal((3, 512)).astype(np.float32)).cuda() if split else None
        for step in range(3):
            if step == 2:
                state[12:16].view(torch.float32).fill_(5e-4)          # a scheduled rate for the third step
            if split:
                jobs = (nv.ReduceJob * 2)()
                out = torch.zeros(512, device="cuda")
                jobs[0] = nv.ReduceJob(P.data_ptr(), None, out.data_ptr(), None, 512, 512, 3, 512, 0, 2)
                second = ctypes.cast(ctypes.addressof(jobs) + ctypes.sizeof(nv.ReduceJob), ctypes.POINTER(nv.ReduceJob))
                nv.check(nv.lib.lidbox_adam_prepare_job(nv.ptr(state), 1e-3, 0.9, 0.999, second))
                nv.check(nv.lib.lidbox_reduce_jobs_run(jobs, 2, st))
                nv.check(nv.lib.lidbox_adam_apply(nv.ptr(pd), nv.ptr(gd), nv.ptr(md), nv.ptr(vd), n, 0.9, 0.999, 1e-7, 0.5, nv.ptr(state), st))
                Pn = P.cpu().numpy()
                assert np.array_equal(out.cpu().numpy(), (Pn[0] + Pn[1]) + Pn[2])
            else:
                nv.check(nv.lib.lidbox_adam_step(nv.ptr(pd), nv.ptr(gd), nv.ptr(md), nv.ptr(vd), n, 1e-3, 0.9, 0.999, 1e-7, 0.5, nv.ptr(state), st))
        torch.cuda.synchronize()
        assert int(state[:8].view(torch.int64).item()) == 3
        res.append((pd.cpu(), md.cpu(), vd.cpu()))
    for a, b in zip(*res):
        assert torch.equal(a, b)
    # a GEMM launch does not carry the optimizer's job
    job = nv.ReduceJob()
    nv.check(nv.lib.lidbox_adam_prepare_job(nv.ptr(state), 1e-3, 0.9, 0.999, ctypes.byref(job)))
    a = torch.zeros((256, 64), device="cuda"); w = torch.zeros((64, 64), device="cuda"); c = torch.zeros((256, 64), device="cuda")
    rc = nv.lib.lidbox_gemm_nt_carry(nv.Rows(a.data_ptr(), 0, 64, 1, 256), nv.ptr(w), 64, nv.Rows(c.data_ptr(), 0, 64, 1, 256), 64, 64, nv.EPI_NONE,
                                     None, None, 0, ctypes.byref(job), 1, st)
    assert rc != 0 and b"optimizer-prepare" in nv.lib.lidbox_hip_last_error()
