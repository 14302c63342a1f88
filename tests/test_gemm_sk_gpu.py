"""
GPU parity of the persistent stream-K fp32 GEMM kernels (csrc/gemm_sk.h) through the C ABI against float64 numpy.

The kernels are the default for large aligned problems (>= 3 GFLOP); the two test aids LIDBOX_GEMM_SK_GRID /
LIDBOX_GEMM_SK_MIN_FLOP shrink the persistent grid and drop the size floor so that SMALL problems walk every branch of the
schedule: whole tiles, streamed tiles with 2 .. many contributors, the in-launch fixed-order reduce, K tails (K % 16 != 0),
ragged M / N edges, implicit-row (conv layout) operands and outputs, every epilogue.  Full-size shapes (BASELINE
configs[1], bs 256) run on the real 768-workgroup grid at the end.
Tolerance: rel 2e-5 of the result scale vs float64 (fp32 round-off of a K-long fmaf chain), as in test_ops_gpu.py.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev(x):
    return torch.from_numpy(np.ascontiguousarray(np.asarray(x, np.float32))).cuda()


def _rows(t, bs, rs, batch, rpb, off_floats=0):
    from lidbox_amd import _native as nv
    return nv.Rows(t.data_ptr() + 4 * off_floats, bs, rs, batch, rpb)


def _close(got, ref, rel=2e-5):
    scale = max(1e-30, float(np.abs(ref).max()))
    err = float(np.abs(got - ref).max())
    assert err <= rel * scale, (err, scale)


def _garbage_ws(nbytes):
    """a workspace that was never initialised: the arrival counters must cope (launch epochs)"""
    ws = torch.empty(max(16, nbytes), dtype=torch.uint8, device="cuda")
    ws.fill_(0xAB)
    return ws


@pytest.fixture
def small_grid(monkeypatch):
    def set_grid(g):
        monkeypatch.setenv("LIDBOX_GEMM_SK_GRID", str(g))
        monkeypatch.setenv("LIDBOX_GEMM_SK_MIN_FLOP", "0")
        monkeypatch.setenv("LIDBOX_GEMM_SK_ALL", "1")
    return set_grid


# grid 16 / 40: whole rounds + a remainder cut into 1 .. 5 parts per tile; 64 / 104: equal spans, every tile split (T < P)
@pytest.mark.parametrize("grid", [16, 40, 64, 104])
@pytest.mark.parametrize("M,K,N", [(1300, 600, 500), (777, 200, 1500), (1408, 1500, 512)])
def test_stream_k_nn_nt_match_float64(grid, M, K, N, small_grid):
    from lidbox_amd import _native as nv
    small_grid(grid)
    rng = np.random.default_rng(M + 3 * K + grid)
    A, Bm, Bt = rng.standard_normal((M, K)), rng.standard_normal((K, N)), rng.standard_normal((N, K))
    bias, mask = rng.standard_normal(N), rng.standard_normal((M, N))
    a, b, bt, bi, mk = _dev(A), _dev(Bm), _dev(Bt), _dev(bias), _dev(mask)
    st = nv.current_stream()
    wsb = nv.lib.lidbox_gemm_rows_workspace(M, N, K)
    if not nv.lib.lidbox_gemm_plan_is_stream_k(0, M, N, K, wsb):
        pytest.skip("spans shorter than 8 K steps at this grid: the planner keeps the classic kernels")
    ws = _garbage_ws(wsb)
    c = torch.full((M, N), 7.0, device="cuda")
    nv.check(nv.lib.lidbox_gemm_nn(_rows(a, 0, K, 1, M), nv.ptr(b), N, _rows(c, 0, N, 1, M), K, N, nv.EPI_BIAS_RELU, nv.ptr(bi),
                                   nv.ptr(ws), wsb, st))
    first = c.clone()
    _close(c.cpu().numpy(), np.maximum(A @ Bm + bias, 0))
    # same launch again: bit-identical (fixed summation order whoever arrives last)
    c.fill_(-1.0)
    nv.check(nv.lib.lidbox_gemm_nn(_rows(a, 0, K, 1, M), nv.ptr(b), N, _rows(c, 0, N, 1, M), K, N, nv.EPI_BIAS_RELU, nv.ptr(bi),
                                   nv.ptr(ws), wsb, st))
    assert torch.equal(c, first)
    c.fill_(-2.0)
    nv.check(nv.lib.lidbox_gemm_nt(_rows(a, 0, K, 1, M), nv.ptr(bt), K, _rows(c, 0, N, 1, M), K, N, nv.EPI_ACCUM_RELU_MASK,
                                   nv.ptr(mk), nv.ptr(ws), wsb, st))
    _close(c.cpu().numpy(), (A @ Bt.T) * (mask > 0) - 2.0)
    nv.check(nv.lib.lidbox_gemm_nt(_rows(a, 0, K, 1, M), nv.ptr(bt), K, _rows(c, 0, N, 1, M), K, N, nv.EPI_RELU_MASK,
                                   nv.ptr(mk), nv.ptr(ws), wsb, st))
    _close(c.cpu().numpy(), (A @ Bt.T) * (mask > 0))
    # the classic kernels on the same inputs: same values to fp32 round-off (different summation order)
    ref = torch.zeros_like(c)
    nv.check(nv.lib.lidbox_gemm_nt(_rows(a, 0, K, 1, M), nv.ptr(bt), K, _rows(ref, 0, N, 1, M), K, N, nv.EPI_RELU_MASK,
                                   nv.ptr(mk), None, 0, st))
    _close(c.cpu().numpy(), ref.cpu().numpy().astype(np.float64))


@pytest.mark.parametrize("grid", [16, 40])
def test_stream_k_conv_layout_rows(grid, small_grid):
    """the conv layout: A rows are overlapping causal windows of a padded [B, Tp, C] activation (k = 3, stride 2), the dgrad
    output rows are strided windows with an utterance gap, masks read through the same descriptor"""
    from lidbox_amd import _native as nv
    small_grid(grid)
    rng = np.random.default_rng(grid)
    Bn, T, C, k, s, Co = 24, 99, 128, 3, 2, 256
    To, Tp = (T - 1) // s + 1, T + k - 1
    X = np.zeros((Bn, Tp, C))
    X[:, k - 1:, :] = rng.standard_normal((Bn, T, C))
    W, bias = rng.standard_normal((k * C, Co)) * 0.1, rng.standard_normal(Co)
    M, K = Bn * To, k * C
    win = np.stack([X[:, t * s:t * s + k, :].reshape(Bn, K) for t in range(To)], axis=1).reshape(M, K)
    x, w, bi = _dev(X), _dev(W), _dev(bias)
    y = torch.full((Bn, To, Co), 5.0, device="cuda")
    st = nv.current_stream()
    wsb = nv.lib.lidbox_gemm_rows_workspace(M, Co, K)
    assert nv.lib.lidbox_gemm_plan_is_stream_k(0, M, Co, K, wsb)
    ws = _garbage_ws(wsb)
    Ad = _rows(x, Tp * C, s * C, Bn, To)
    nv.check(nv.lib.lidbox_gemm_nn(Ad, nv.ptr(w), Co, _rows(y, To * Co, Co, Bn, To), K, Co, nv.EPI_BIAS_RELU, nv.ptr(bi),
                                   nv.ptr(ws), wsb, st))
    ref_y = np.maximum(win @ W + bias, 0)
    _close(y.cpu().numpy().reshape(M, Co), ref_y)
    # dgrad of tap group 0 (taps 0..s-1): dX windows [t*s, t*s + s) += mask * (dY . W[:s*C]^T), rows strided by s*C
    dY = rng.standard_normal((M, Co))
    dy = _dev(dY)
    dx = torch.zeros((Bn, Tp, C), device="cuda")
    Kd, Nd = Co, s * C
    wsb2 = nv.lib.lidbox_gemm_rows_workspace(M, Nd, Kd)
    ws2 = _garbage_ws(wsb2)
    Cd = _rows(dx, Tp * C, s * C, Bn, To)
    nv.check(nv.lib.lidbox_gemm_nt(_rows(dy, 0, Co, 1, M), nv.ptr(w), Co, Cd, Kd, Nd, nv.EPI_RELU_MASK, nv.ptr(x), nv.ptr(ws2), wsb2, st))
    got = dx.cpu().numpy()
    full = dY @ W[:Nd].T                                          # [M, s*C]
    ref = np.zeros((Bn, Tp, C))
    for b_ in range(Bn):
        for t in range(To):
            blk = full[b_ * To + t].reshape(s, C)
            rows = X[b_, t * s:t * s + s, :]
            ref[b_, t * s:t * s + s, :] = blk * (rows > 0)
    _close(got, ref)


@pytest.mark.parametrize("grid", [8, 40, 256])
@pytest.mark.parametrize("M,K1,N", [(5000, 200, 500), (3333, 512, 1500)])
def test_stream_k_body_wgrad_matches_float64(grid, M, K1, N, small_grid):
    from lidbox_amd import _native as nv
    small_grid(grid)
    rng = np.random.default_rng(M + grid)
    A, Bm = rng.standard_normal((M, K1)), rng.standard_normal((M, N))
    a, b = _dev(A), _dev(Bm)
    st = nv.current_stream()
    wsb = nv.lib.lidbox_gemm_tn_workspace(M, K1, N)
    if not nv.lib.lidbox_gemm_plan_is_stream_k(2, M, N, K1, wsb):
        pytest.skip("slices shorter than 128 rows at this grid")
    ws = _garbage_ws(wsb)
    c = torch.full((K1, N), 3.0, device="cuda")
    g = torch.full((N,), -1.0, device="cuda")
    nv.check(nv.lib.lidbox_gemm_tn(_rows(a, 0, K1, 1, M), _rows(b, 0, N, 1, M), nv.ptr(c), N, K1, N, 0, nv.ptr(g), nv.ptr(ws), wsb, st))
    _close(c.cpu().numpy(), A.T @ Bm)
    _close(g.cpu().numpy(), Bm.sum(0), rel=1e-5)
    nv.check(nv.lib.lidbox_gemm_tn(_rows(a, 0, K1, 1, M), _rows(b, 0, N, 1, M), nv.ptr(c), N, K1, N, 1, nv.ptr(g), nv.ptr(ws), wsb, st))
    _close(c.cpu().numpy(), 2 * (A.T @ Bm))
    _close(g.cpu().numpy(), 2 * Bm.sum(0), rel=1e-5)


def test_stream_k_body_wgrad_conv_layout(small_grid):
    """wgrad whose A rows are the causal windows of a padded activation (utterance wrap inside the contraction)"""
    from lidbox_amd import _native as nv
    small_grid(16)
    rng = np.random.default_rng(5)
    Bn, T, C, k, s, Co = 40, 50, 64, 3, 1, 128
    To, Tp = T, T + k - 1
    X = np.zeros((Bn, Tp, C))
    X[:, k - 1:, :] = rng.standard_normal((Bn, T, C))
    M, K = Bn * To, k * C
    win = np.stack([X[:, t:t + k, :].reshape(Bn, K) for t in range(To)], axis=1).reshape(M, K)
    dY = rng.standard_normal((M, Co))
    x, dy = _dev(X), _dev(dY)
    dw = torch.zeros((K, Co), device="cuda")
    g = torch.zeros((Co,), device="cuda")
    st = nv.current_stream()
    wsb = nv.lib.lidbox_gemm_tn_workspace(M, K, Co)
    assert nv.lib.lidbox_gemm_plan_is_stream_k(2, M, Co, K, wsb)
    ws = _garbage_ws(wsb)
    nv.check(nv.lib.lidbox_gemm_tn(_rows(x, Tp * C, s * C, Bn, To), _rows(dy, 0, Co, 1, M), nv.ptr(dw), Co, K, Co, 0, nv.ptr(g),
                                   nv.ptr(ws), wsb, st))
    _close(dw.cpu().numpy(), win.T @ dY)
    _close(g.cpu().numpy(), dY.sum(0), rel=1e-5)


@pytest.mark.parametrize("kind,M,K,N", [("nn", 50688, 200, 512), ("nn", 8448, 512, 1500), ("nt", 8448, 1500, 512),
                                         ("nt", 25344, 512, 1024), ("nn", 25344, 1536, 512), ("tn", 8448, 512, 1500), ("tn", 50688, 200, 512)])
def test_stream_k_full_size_layers_on_the_real_grid(kind, M, K, N, monkeypatch):
    """x-vector layer shapes at bs 256 (BASELINE configs[1]) on the production 768-workgroup grid vs torch float64 on the GPU
    (LIDBOX_GEMM_SK_ALL: also the shapes the default policy leaves to the classic kernels)"""
    from lidbox_amd import _native as nv
    monkeypatch.setenv("LIDBOX_GEMM_SK_ALL", "1")
    gen = torch.Generator(device="cuda").manual_seed(M + K)
    st = nv.current_stream()
    if kind == "tn":
        a = torch.randn(M, K, device="cuda", generator=gen)
        b = torch.randn(M, N, device="cuda", generator=gen)
        wsb = nv.lib.lidbox_gemm_tn_workspace(M, K, N)
        assert nv.lib.lidbox_gemm_plan_is_stream_k(2, M, N, K, wsb)
        ws = _garbage_ws(wsb)
        c = torch.zeros(K, N, device="cuda")
        g = torch.zeros(N, device="cuda")
        nv.check(nv.lib.lidbox_gemm_tn(_rows(a, 0, K, 1, M), _rows(b, 0, N, 1, M), nv.ptr(c), N, K, N, 0, nv.ptr(g), nv.ptr(ws), wsb, st))
        ref = a.double().t() @ b.double()
        assert float((c.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
        refg = b.double().sum(0)
        assert float((g.double() - refg).abs().max()) <= 1e-5 * float(refg.abs().max())
        return
    a = torch.randn(M, K, device="cuda", generator=gen)
    wsb = nv.lib.lidbox_gemm_rows_workspace(M, N, K)
    assert nv.lib.lidbox_gemm_plan_is_stream_k(0 if kind == "nn" else 1, M, N, K, wsb)
    ws = _garbage_ws(wsb)
    c = torch.zeros(M, N, device="cuda")
    if kind == "nn":
        b = torch.randn(K, N, device="cuda", generator=gen)
        bias = torch.randn(N, device="cuda", generator=gen)
        nv.check(nv.lib.lidbox_gemm_nn(_rows(a, 0, K, 1, M), nv.ptr(b), N, _rows(c, 0, N, 1, M), K, N, nv.EPI_BIAS_RELU, nv.ptr(bias),
                                       nv.ptr(ws), wsb, st))
        ref = torch.relu(a.double() @ b.double() + bias.double())
    else:
        b = torch.randn(N, K, device="cuda", generator=gen)
        mask = torch.randn(M, N, device="cuda", generator=gen)
        nv.check(nv.lib.lidbox_gemm_nt(_rows(a, 0, K, 1, M), nv.ptr(b), K, _rows(c, 0, N, 1, M), K, N, nv.EPI_RELU_MASK, nv.ptr(mask),
                                       nv.ptr(ws), wsb, st))
        ref = (a.double() @ b.double().t()) * (mask > 0)
    assert float((c.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())


@pytest.mark.parametrize("grid", [16, 40, 64])
def test_pipelined_epilogue_variant_matches_float64(grid, small_grid, monkeypatch):
    """gemm_skp_rows_kernel (LIDBOX_GEMM_SKP=2; measured, not adopted: profiles/r03_gemm_skp_ab.txt): a finished tile's epilogue
    -- bias / ReLU, ReLU mask, accumulate -- is drained behind the next item's MFMA groups, the mask and old values staged by
    LDS-DMA.  Same answers as float64 through whole rounds, remainder parts and equal spans, plain and conv-layout outputs
    (utterance gap inside a 32-row block), bit-identical run to run."""
    from lidbox_amd import _native as nv
    small_grid(grid)
    monkeypatch.setenv("LIDBOX_GEMM_SKP", "2")
    rng = np.random.default_rng(100 + grid)
    M, K, N = 1300, 600, 500
    A, Bm, Bt = rng.standard_normal((M, K)), rng.standard_normal((K, N)), rng.standard_normal((N, K))
    bias, mask = rng.standard_normal(N), rng.standard_normal((M, N))
    a, b, bt, bi, mk = _dev(A), _dev(Bm), _dev(Bt), _dev(bias), _dev(mask)
    st = nv.current_stream()
    wsb = nv.lib.lidbox_gemm_rows_workspace(M, N, K)
    ws = _garbage_ws(wsb)
    c = torch.full((M, N), 7.0, device="cuda")
    nv.check(nv.lib.lidbox_gemm_nn(_rows(a, 0, K, 1, M), nv.ptr(b), N, _rows(c, 0, N, 1, M), K, N, nv.EPI_BIAS_RELU, nv.ptr(bi),
                                   nv.ptr(ws), wsb, st))
    first = c.clone()
    _close(c.cpu().numpy(), np.maximum(A @ Bm + bias, 0))
    c.fill_(-1.0)
    nv.check(nv.lib.lidbox_gemm_nn(_rows(a, 0, K, 1, M), nv.ptr(b), N, _rows(c, 0, N, 1, M), K, N, nv.EPI_BIAS_RELU, nv.ptr(bi),
                                   nv.ptr(ws), wsb, st))
    assert torch.equal(c, first)
    c.fill_(-2.0)
    nv.check(nv.lib.lidbox_gemm_nt(_rows(a, 0, K, 1, M), nv.ptr(bt), K, _rows(c, 0, N, 1, M), K, N, nv.EPI_ACCUM_RELU_MASK,
                                   nv.ptr(mk), nv.ptr(ws), wsb, st))
    _close(c.cpu().numpy(), (A @ Bt.T) * (mask > 0) - 2.0)
    nv.check(nv.lib.lidbox_gemm_nt(_rows(a, 0, K, 1, M), nv.ptr(bt), K, _rows(c, 0, N, 1, M), K, N, nv.EPI_RELU_MASK,
                                   nv.ptr(mk), nv.ptr(ws), wsb, st))
    _close(c.cpu().numpy(), (A @ Bt.T) * (mask > 0))
    # conv layout: 26 utterances of 50 output rows inside [26, 53, N] buffers (3 leading rows per utterance untouched)
    Bn, R, Rp = 26, 50, 53
    cb = torch.full((Bn, Rp, N), 7.0, device="cuda")
    mkb = torch.zeros((Bn, Rp, N), device="cuda")
    mkb[:, 3:, :] = mk[:Bn * R].reshape(Bn, R, N)
    cv, mv = cb[:, 3:, :], mkb[:, 3:, :]
    Cd = nv.Rows(cv.data_ptr(), Rp * N, N, Bn, R)
    nv.check(nv.lib.lidbox_gemm_nt(_rows(a, 0, K, 1, Bn * R), nv.ptr(bt), K, Cd, K, N, nv.EPI_ACCUM_RELU_MASK,
                                   nv.C.c_void_p(mv.data_ptr()), nv.ptr(ws), wsb, st))
    got = cb.cpu().numpy()
    ref = (A[:Bn * R] @ Bt.T) * (mask[:Bn * R] > 0) + 7.0
    _close(got[:, 3:, :].reshape(Bn * R, N), ref)
    assert np.all(got[:, :3, :] == 7.0)


# ---- streamed remainder of the one-tile-per-workgroup LDS-DMA kernels (csrc/gemm_dma.h: DmaStream) ----------------------
# tiles % 256 tiles of the last partial round are cut along K into g pieces; the last arriver sums the slabs in k order.
@pytest.mark.parametrize("plan,M,N", [("64,64,1", 64 * 70 + 13, 256), ("128,64,1", 128 * 69 - 5, 250), ("64,128,1", 64 * 41, 896),
                                      ("128,128,1", 128 * 66, 512)])
@pytest.mark.parametrize("K,g", [(600, 0), (200, 2), (1536, 8), (512, 3)])
def test_streamed_remainder_matches_float64(plan, M, N, K, g, monkeypatch):
    """every tile shape, K tails (600 = 37.5 steps), ragged M / N edges in the streamed tiles (they are the LAST tiles: the
    bottom rows), planner-chosen and forced piece counts, every epilogue, a garbage workspace, bit-identical replays"""
    from lidbox_amd import _native as nv
    monkeypatch.setenv("LIDBOX_GEMM_PLAN", plan)
    if g:
        monkeypatch.setenv("LIDBOX_GEMM_STREAM_TAIL", str(g))
    rng = np.random.default_rng(M + K + g)
    A, Bm, Bt = rng.standard_normal((M, K)), rng.standard_normal((K, N)), rng.standard_normal((N, K))
    bias, mask = rng.standard_normal(N), rng.standard_normal((M, N))
    a, b, bt, bi, mk = _dev(A), _dev(Bm), _dev(Bt), _dev(bias), _dev(mask)
    st = nv.current_stream()
    wsb = nv.lib.lidbox_gemm_rows_workspace(M, N, K) + (64 << 20)
    pieces = nv.lib.lidbox_gemm_plan_stream_tail(0, M, N, K, wsb)
    if g:
        assert pieces == g
    elif pieces == 0:
        pytest.skip("the planner streams nothing here (remainder close to a whole round, or K too short)")
    ws = _garbage_ws(wsb)
    if N % 4 == 0:                                               # nn needs 16-byte aligned B rows for the DMA family
        c = torch.full((M, N), 7.0, device="cuda")
        nv.check(nv.lib.lidbox_gemm_nn(_rows(a, 0, K, 1, M), nv.ptr(b), N, _rows(c, 0, N, 1, M), K, N, nv.EPI_BIAS_RELU, nv.ptr(bi),
                                       nv.ptr(ws), wsb, st))
        assert nv.lib.lidbox_gemm_last_family() == 1
        first = c.clone()
        _close(c.cpu().numpy(), np.maximum(A @ Bm + bias, 0))
        c.fill_(-1.0)
        nv.check(nv.lib.lidbox_gemm_nn(_rows(a, 0, K, 1, M), nv.ptr(b), N, _rows(c, 0, N, 1, M), K, N, nv.EPI_BIAS_RELU, nv.ptr(bi),
                                       nv.ptr(ws), wsb, st))
        assert torch.equal(c, first)
    c = torch.full((M, N), -2.0, device="cuda")
    nv.check(nv.lib.lidbox_gemm_nt(_rows(a, 0, K, 1, M), nv.ptr(bt), K, _rows(c, 0, N, 1, M), K, N, nv.EPI_ACCUM_RELU_MASK,
                                   nv.ptr(mk), nv.ptr(ws), wsb, st))
    assert nv.lib.lidbox_gemm_last_family() == 1
    _close(c.cpu().numpy(), (A @ Bt.T) * (mask > 0) - 2.0)
    nv.check(nv.lib.lidbox_gemm_nt(_rows(a, 0, K, 1, M), nv.ptr(bt), K, _rows(c, 0, N, 1, M), K, N, nv.EPI_RELU_MASK,
                                   nv.ptr(mk), nv.ptr(ws), wsb, st))
    first = c.clone()
    _close(c.cpu().numpy(), (A @ Bt.T) * (mask > 0))
    # unstreamed launch of the same decomposition: same values to fp32 round-off
    monkeypatch.setenv("LIDBOX_GEMM_STREAM_TAIL", "0")
    assert nv.lib.lidbox_gemm_plan_stream_tail(1, M, N, K, wsb) == 0
    ref = torch.zeros_like(c)
    nv.check(nv.lib.lidbox_gemm_nt(_rows(a, 0, K, 1, M), nv.ptr(bt), K, _rows(ref, 0, N, 1, M), K, N, nv.EPI_RELU_MASK,
                                   nv.ptr(mk), nv.ptr(ws), wsb, st))
    _close(c.cpu().numpy(), ref.cpu().numpy().astype(np.float64))
    # 50 replays on the same workspace (arrival order varies): bit-identical
    monkeypatch.setenv("LIDBOX_GEMM_STREAM_TAIL", str(g) if g else "1")
    for _ in range(50):
        nv.check(nv.lib.lidbox_gemm_nt(_rows(a, 0, K, 1, M), nv.ptr(bt), K, _rows(c, 0, N, 1, M), K, N, nv.EPI_RELU_MASK,
                                       nv.ptr(mk), nv.ptr(ws), wsb, st))
    assert torch.equal(c, first)


def test_streamed_remainder_conv_layout_and_small_workspace(monkeypatch):
    """conv-layout operands (overlapping windows in, strided windows with an utterance gap out) through the streamed tiles;
    a workspace smaller than the slabs need falls back to whole tiles (same values)"""
    from lidbox_amd import _native as nv
    monkeypatch.setenv("LIDBOX_GEMM_PLAN", "64,64,1")
    rng = np.random.default_rng(5)
    Bn, T, C, k, s, Co = 96, 99, 128, 3, 2, 256
    To, Tp = (T - 1) // s + 1, T + k - 1
    X = np.zeros((Bn, Tp, C))
    X[:, k - 1:, :] = rng.standard_normal((Bn, T, C))
    W, bias = rng.standard_normal((k * C, Co)) * 0.1, rng.standard_normal(Co)
    M, K = Bn * To, k * C                                        # 4800 x 384: 75 x 4 = 300 tiles, 44 streamed
    win = np.stack([X[:, t * s:t * s + k, :].reshape(Bn, K) for t in range(To)], axis=1).reshape(M, K)
    x, w, bi = _dev(X), _dev(W), _dev(bias)
    y = torch.full((Bn, To, Co), 5.0, device="cuda")
    st = nv.current_stream()
    wsb = nv.lib.lidbox_gemm_rows_workspace(M, Co, K)
    assert nv.lib.lidbox_gemm_plan_stream_tail(0, M, Co, K, wsb) >= 2
    ws = _garbage_ws(wsb)
    Ad = _rows(x, Tp * C, s * C, Bn, To)
    nv.check(nv.lib.lidbox_gemm_nn(Ad, nv.ptr(w), Co, _rows(y, To * Co, Co, Bn, To), K, Co, nv.EPI_BIAS_RELU, nv.ptr(bi),
                                   nv.ptr(ws), wsb, st))
    _close(y.cpu().numpy().reshape(M, Co), np.maximum(win @ W + bias, 0))
    assert nv.lib.lidbox_gemm_plan_stream_tail(0, M, Co, K, 20000) == 0
    y2 = torch.full((Bn, To, Co), 5.0, device="cuda")
    nv.check(nv.lib.lidbox_gemm_nn(Ad, nv.ptr(w), Co, _rows(y2, To * Co, Co, Bn, To), K, Co, nv.EPI_BIAS_RELU, nv.ptr(bi),
                                   nv.ptr(ws), 20000, st))
    _close(y2.cpu().numpy().reshape(M, Co), y.cpu().numpy().reshape(M, Co).astype(np.float64))
    dY = rng.standard_normal((M, Co))
    dy = _dev(dY)
    dx = torch.zeros((Bn, Tp, C), device="cuda")
    Kd, Nd = Co, s * C
    wsb2 = nv.lib.lidbox_gemm_rows_workspace(M, Nd, Kd)
    assert nv.lib.lidbox_gemm_plan_stream_tail(1, M, Nd, Kd, wsb2) >= 2
    ws2 = _garbage_ws(wsb2)
    Cd = _rows(dx, Tp * C, s * C, Bn, To)
    nv.check(nv.lib.lidbox_gemm_nt(_rows(dy, 0, Co, 1, M), nv.ptr(w), Co, Cd, Kd, Nd, nv.EPI_RELU_MASK, nv.ptr(x), nv.ptr(ws2), wsb2, st))
    full = dY @ W[:Nd].T
    ref = np.zeros((Bn, Tp, C))
    for b_ in range(Bn):
        for t in range(To):
            ref[b_, t * s:t * s + s, :] = full[b_ * To + t].reshape(s, C) * (X[b_, t * s:t * s + s, :] > 0)
    _close(dx.cpu().numpy(), ref)


# ---- a layer's dgrad + wgrad as one launch (lidbox_gemm_nt_tn, csrc/gemm_dma.h: gemm_nt_tn_pair_kernel) ------------------
@pytest.mark.parametrize("M,Co,N,K1,epi", [(256, 512, 3000, 3000, "none"), (256, 512, 512, 512, "mask"), (200, 64, 100, 100, "mask"),
                                           (8448, 512, 512, 512, "mask")])
def test_dgrad_wgrad_pair_is_bit_identical_to_the_two_calls(M, Co, N, K1, epi, monkeypatch):
    """the dense head's shapes (M = 256 batch rows: 3000 -> 512 -> 512) go out as one kernel, ragged shapes too; a conv-size
    problem falls back to the two launches; either way the results equal lidbox_gemm_tn + lidbox_gemm_nt bit for bit, and
    float64 to round-off"""
    from lidbox_amd import _native as nv
    rng = np.random.default_rng(M + N)
    dY, W, X = rng.standard_normal((M, Co)), rng.standard_normal((N, Co)) * 0.1, rng.standard_normal((M, K1))
    mask = rng.standard_normal((M, N))
    dy, w, x, mk = _dev(dY), _dev(W), _dev(X), _dev(mask)
    st = nv.current_stream()
    ws1 = _garbage_ws(max(nv.lib.lidbox_gemm_rows_workspace(M, N, Co), 16) + 1024)
    ws2 = _garbage_ws(max(nv.lib.lidbox_gemm_tn_workspace(M, K1, Co), 16) + 1024)
    e = nv.EPI_RELU_MASK if epi == "mask" else nv.EPI_NONE
    aux = nv.ptr(mk) if epi == "mask" else None
    def run(pair):
        dx = torch.full((M, N), 9.0, device="cuda"); dw = torch.full((K1, Co), 9.0, device="cuda"); db = torch.full((Co,), 9.0, device="cuda")
        if pair:
            nv.check(nv.lib.lidbox_gemm_nt_tn(_rows(dy, 0, Co, 1, M), nv.ptr(w), Co, _rows(dx, 0, N, 1, M), Co, N, e, aux, nv.ptr(ws1),
                                              ws1.numel(), _rows(x, 0, K1, 1, M), nv.ptr(dw), Co, K1, 0, nv.ptr(db), nv.ptr(ws2), ws2.numel(), st))
            launches = (nv.C.c_int * 3)()
            nv.check(nv.lib.lidbox_gemm_last_launches(launches))
        else:
            nv.check(nv.lib.lidbox_gemm_tn(_rows(x, 0, K1, 1, M), _rows(dy, 0, Co, 1, M), nv.ptr(dw), Co, K1, Co, 0, nv.ptr(db), nv.ptr(ws2),
                                           ws2.numel(), st))
            nv.check(nv.lib.lidbox_gemm_nt(_rows(dy, 0, Co, 1, M), nv.ptr(w), Co, _rows(dx, 0, N, 1, M), Co, N, e, aux, nv.ptr(ws1), ws1.numel(), st))
        return dx, dw, db
    a, b = run(True), run(False)
    for u, v in zip(a, b):
        assert torch.equal(u, v)
    ref_dx = dY @ W.T
    if epi == "mask":
        ref_dx = ref_dx * (mask > 0)
    _close(a[0].cpu().numpy(), ref_dx)
    _close(a[1].cpu().numpy(), X.T @ dY)
    _close(a[2].cpu().numpy(), dY.sum(0), rel=1e-5)
    monkeypatch.setenv("LIDBOX_GEMM_NO_PAIR", "1")
    c = run(True)
    for u, v in zip(a, c):
        assert torch.equal(u, v)


def test_dgrad_wgrad_pair_shared_workspace_and_bad_arguments():
    """one workspace for both halves: the call must not run them concurrently (it falls back to wgrad, then dgrad -- same
    values); invalid descriptors are rejected with the error text of the underlying entry points"""
    from lidbox_amd import _native as nv
    rng = np.random.default_rng(3)
    M, Co, N, K1 = 256, 512, 512, 512
    dY, W, X = rng.standard_normal((M, Co)), rng.standard_normal((N, Co)) * 0.1, rng.standard_normal((M, K1))
    dy, w, x = _dev(dY), _dev(W), _dev(X)
    st = nv.current_stream()
    ws = _garbage_ws(max(nv.lib.lidbox_gemm_rows_workspace(M, N, Co), nv.lib.lidbox_gemm_tn_workspace(M, K1, Co)) + 1024)
    dx = torch.zeros((M, N), device="cuda"); dw = torch.zeros((K1, Co), device="cuda"); db = torch.zeros((Co,), device="cuda")
    nv.check(nv.lib.lidbox_gemm_nt_tn(_rows(dy, 0, Co, 1, M), nv.ptr(w), Co, _rows(dx, 0, N, 1, M), Co, N, nv.EPI_NONE, None, nv.ptr(ws),
                                      ws.numel(), _rows(x, 0, K1, 1, M), nv.ptr(dw), Co, K1, 0, nv.ptr(db), nv.ptr(ws), ws.numel(), st))
    launches = (nv.C.c_int * 3)()
    nv.check(nv.lib.lidbox_gemm_last_launches(launches))
    _close(dx.cpu().numpy(), dY @ W.T)
    _close(dw.cpu().numpy(), X.T @ dY)
    _close(db.cpu().numpy(), dY.sum(0), rel=1e-5)
    # row counts of X and dY differ
    rc = nv.lib.lidbox_gemm_nt_tn(_rows(dy, 0, Co, 1, M), nv.ptr(w), Co, _rows(dx, 0, N, 1, M), Co, N, nv.EPI_NONE, None, nv.ptr(ws), ws.numel(),
                                  _rows(x, 0, K1, 1, M - 1), nv.ptr(dw), Co, K1, 0, nv.ptr(db), None, 0, st)
    assert rc != 0 and b"row counts" in nv.lib.lidbox_hip_last_error()
    # a mask epilogue without its mask
    rc = nv.lib.lidbox_gemm_nt_tn(_rows(dy, 0, Co, 1, M), nv.ptr(w), Co, _rows(dx, 0, N, 1, M), Co, N, nv.EPI_RELU_MASK, None, nv.ptr(ws),
                                  ws.numel(), _rows(x, 0, K1, 1, M), nv.ptr(dw), Co, K1, 0, nv.ptr(db), nv.ptr(ws), ws.numel(), st)
    assert rc != 0 and b"aux" in nv.lib.lidbox_hip_last_error()


@pytest.mark.parametrize("M,Co,N,K1,epi,accumulate,bias", [
    (8448, 512, 512, 1536, "mask", 0, True),       # frame3 at bs 256: whole rounds + a streamed remainder behind the carried blocks
    (8448, 1500, 512, 512, "none", 0, True),       # frame5: K tail (1500), 64 x 64 tiles
    (4100, 512, 1024, 512, "mask", 1, False),      # ragged rows, accumulate into dW, no bias gradient
    (700, 256, 192, 200, "none", 0, True),         # fewer tiles than one round; K1 = 200 (frame1's width)
])
def test_carried_reduce_is_bit_identical_to_the_separate_launches(M, Co, N, K1, epi, accumulate, bias, monkeypatch):
    """lidbox_gemm_tn_partial + lidbox_gemm_nt_carry (the wgrad's fixed-order slice sum in the leading workgroups of the dgrad
    launch) == lidbox_gemm_tn + lidbox_gemm_nt bit for bit, == the job run on its own, == LIDBOX_GEMM_NO_CARRY=1; float64 to
    round-off.  Reference shapes: lidbox/models/xvector.py:53-57 backward."""
    from lidbox_amd import _native as nv
    rng = np.random.default_rng(M + N + K1)
    dY, W, X = rng.standard_normal((M, Co)), rng.standard_normal((N, Co)) * 0.1, rng.standard_normal((M, K1))
    mask, dW0 = rng.standard_normal((M, N)), rng.standard_normal((K1, Co))
    dy, w, x, mk = _dev(dY), _dev(W), _dev(X), _dev(mask)
    st = nv.current_stream()
    ws1 = _garbage_ws(max(nv.lib.lidbox_gemm_rows_workspace(M, N, Co), 16) + 1024)
    ws2 = _garbage_ws(max(nv.lib.lidbox_gemm_tn_workspace(M, K1, Co), 16) + 1024)
    e = nv.EPI_RELU_MASK if epi == "mask" else nv.EPI_NONE
    aux = nv.ptr(mk) if epi == "mask" else None

    def run(mode):
        dx = torch.full((M, N), 9.0, device="cuda")
        dw = _dev(dW0).clone()
        db = torch.full((Co,), 9.0, device="cuda") if bias else None
        A, Bd, Cx = _rows(x, 0, K1, 1, M), _rows(dy, 0, Co, 1, M), _rows(dx, 0, N, 1, M)
        carried = None
        if mode == "plain":
            nv.check(nv.lib.lidbox_gemm_tn(A, Bd, nv.ptr(dw), Co, K1, Co, accumulate, nv.ptr(db), nv.ptr(ws2), ws2.numel(), st))
            nv.check(nv.lib.lidbox_gemm_nt(Bd, nv.ptr(w), Co, Cx, Co, N, e, aux, nv.ptr(ws1), ws1.numel(), st))
        else:
            job = nv.ReduceJob()
            nv.check(nv.lib.lidbox_gemm_tn_partial(A, Bd, nv.ptr(dw), Co, K1, Co, accumulate, nv.ptr(db), nv.ptr(ws2), ws2.numel(),
                                                   nv.C.byref(job), st))
            assert job.nblocks > 0 and job.splits >= 1 and job.n == K1 * Co
            if mode == "job_alone":
                nv.check(nv.lib.lidbox_reduce_jobs_run(nv.C.byref(job), 1, st))
                nv.check(nv.lib.lidbox_gemm_nt_carry(Bd, nv.ptr(w), Co, Cx, Co, N, e, aux, nv.ptr(ws1), ws1.numel(), None, 0, st))
            else:
                nv.check(nv.lib.lidbox_gemm_nt_carry(Bd, nv.ptr(w), Co, Cx, Co, N, e, aux, nv.ptr(ws1), ws1.numel(), nv.C.byref(job), 1, st))
                carried = nv.lib.lidbox_gemm_last_carried()
        torch.cuda.synchronize()
        return (dx, dw) + ((db,) if bias else ()), carried

    ref, _ = run("plain")
    got, carried = run("carry")
    assert carried == 1, "16-byte aligned operands: the reduce must ride in the dgrad launch"
    for u, v in zip(got, ref):
        assert torch.equal(u, v)
    alone, _ = run("job_alone")
    for u, v in zip(alone, ref):
        assert torch.equal(u, v)
    monkeypatch.setenv("LIDBOX_GEMM_NO_CARRY", "1")
    off, carried = run("carry")
    assert carried == 0
    for u, v in zip(off, ref):
        assert torch.equal(u, v)
    monkeypatch.delenv("LIDBOX_GEMM_NO_CARRY")
    monkeypatch.setenv("LIDBOX_GEMM_CARRY_BLOCKS", "8")            # a handful of blocks walk the whole job
    few, carried = run("carry")
    assert carried == 1
    for u, v in zip(few, ref):
        assert torch.equal(u, v)
    ref_dx = dY @ W.T
    if epi == "mask":
        ref_dx = ref_dx * (mask > 0)
    _close(got[0].cpu().numpy(), ref_dx)
    _close(got[1].cpu().numpy(), X.T @ dY + (dW0 if accumulate else 0.0))
    if bias:
        _close(got[2].cpu().numpy(), dY.sum(0), rel=1e-5)


def test_carried_reduce_conv_layout_and_rejections():
    """the dgrad writes a strided, batched activation-gradient buffer (frame3's layout: k 3, s 3) while it carries the reduce;
    a job whose slices sit in the dgrad's own workspace is refused"""
    from lidbox_amd import _native as nv
    rng = np.random.default_rng(5)
    B, To, Co, cin, k, s = 64, 33, 512, 512, 3, 3
    Tp = To * s + 2
    M, N, K1 = B * To, k * cin, k * cin
    dY, W = rng.standard_normal((M, Co)), rng.standard_normal((N, Co)) * 0.1
    act = rng.standard_normal((B, Tp, cin))
    dy, w, a = _dev(dY), _dev(W), _dev(act)
    st = nv.current_stream()
    ws1 = _garbage_ws(max(nv.lib.lidbox_gemm_rows_workspace(M, N, Co), 16) + 1024)
    ws2 = _garbage_ws(max(nv.lib.lidbox_gemm_tn_workspace(M, K1, Co), 16) + 1024)
    A = _rows(a, Tp * cin, s * cin, B, To)                      # implicit rows: window t = padded rows [t*s, t*s + k)
    outs = []
    for carry in (False, True):
        dprev = torch.zeros((B, Tp, cin), device="cuda")
        dw = torch.zeros((K1, Co), device="cuda"); db = torch.zeros((Co,), device="cuda")
        Cd = _rows(dprev, Tp * cin, s * cin, B, To)
        if carry:
            job = nv.ReduceJob()
            nv.check(nv.lib.lidbox_gemm_tn_partial(A, _rows(dy, 0, Co, 1, M), nv.ptr(dw), Co, K1, Co, 0, nv.ptr(db), nv.ptr(ws2), ws2.numel(),
                                                   nv.C.byref(job), st))
            rc = nv.lib.lidbox_gemm_nt_carry(_rows(dy, 0, Co, 1, M), nv.ptr(w), Co, Cd, Co, N, nv.EPI_RELU_MASK, nv.ptr(a), nv.ptr(ws2),
                                             ws2.numel(), nv.C.byref(job), 1, st)
            assert rc != 0 and b"workspace" in nv.lib.lidbox_hip_last_error()
            nv.check(nv.lib.lidbox_gemm_nt_carry(_rows(dy, 0, Co, 1, M), nv.ptr(w), Co, Cd, Co, N, nv.EPI_RELU_MASK, nv.ptr(a), nv.ptr(ws1),
                                                 ws1.numel(), nv.C.byref(job), 1, st))
            assert nv.lib.lidbox_gemm_last_carried() == 1
        else:
            nv.check(nv.lib.lidbox_gemm_tn(A, _rows(dy, 0, Co, 1, M), nv.ptr(dw), Co, K1, Co, 0, nv.ptr(db), nv.ptr(ws2), ws2.numel(), st))
            nv.check(nv.lib.lidbox_gemm_nt(_rows(dy, 0, Co, 1, M), nv.ptr(w), Co, Cd, Co, N, nv.EPI_RELU_MASK, nv.ptr(a), nv.ptr(ws1),
                                           ws1.numel(), st))
        outs.append((dprev, dw, db))
    for u, v in zip(outs[0], outs[1]):
        assert torch.equal(u, v)
    cols = np.stack([act[:, t * s:t * s + k, :].reshape(B, k * cin) for t in range(To)], axis=1).reshape(M, K1)
    _close(outs[1][1].cpu().numpy(), cols.T @ dY)
    ref = (dY @ W.T).reshape(B, To, k, cin)
    full = np.zeros((B, Tp, cin))
    for t in range(To):
        full[:, t * s:t * s + k, :] = ref[:, t] * (act[:, t * s:t * s + k, :] > 0)
    _close(outs[1][0].cpu().numpy(), full)


def test_pair_launch_hands_its_reduce_on_and_carries_an_earlier_one(monkeypatch):
    """the dense head under the engine: segment2's pair launch leaves its wgrad reduce as a job, segment1's pair launch
    carries it in its leading workgroups and leaves its own, which a conv-size dgrad then carries together with that
    layer's reduce (two jobs in one launch); everything equals the plain sequence of calls bit for bit"""
    from lidbox_amd import _native as nv
    rng = np.random.default_rng(17)
    st = nv.current_stream()
    shapes = [(256, 512, 512, 512), (256, 512, 3000, 3000), (8448, 1500, 512, 512)]          # (M, Co, N, K1): segment2, segment1, frame5
    data = []
    for M, Co, N, K1 in shapes:
        data.append(tuple(_dev(x) for x in (rng.standard_normal((M, Co)), rng.standard_normal((N, Co)) * 0.1, rng.standard_normal((M, K1)))))
    ws_nt = _garbage_ws(max(nv.lib.lidbox_gemm_rows_workspace(M, N, Co) for M, Co, N, K1 in shapes) + 1024)
    need_tn = max(nv.lib.lidbox_gemm_tn_workspace(M, K1, Co) for M, Co, N, K1 in shapes) + 1024
    regions = [_garbage_ws(need_tn), _garbage_ws(need_tn)]

    def run(carry):
        outs, pending = [], []
        for li, ((M, Co, N, K1), (dy, w, x)) in enumerate(zip(shapes, data)):
            dx = torch.full((M, N), 9.0, device="cuda"); dw = torch.full((K1, Co), 9.0, device="cuda"); db = torch.full((Co,), 9.0, device="cuda")
            args = (_rows(dy, 0, Co, 1, M), nv.ptr(w), Co, _rows(dx, 0, N, 1, M), Co, N, nv.EPI_NONE, None, nv.ptr(ws_nt), ws_nt.numel(),
                    _rows(x, 0, K1, 1, M), nv.ptr(dw), Co, K1, 0, nv.ptr(db), nv.ptr(regions[li % 2]), regions[li % 2].numel())
            if carry:
                jobs = (nv.ReduceJob * 1)(*pending)
                out = nv.ReduceJob()
                nv.check(nv.lib.lidbox_gemm_nt_tn_carry(*args, jobs, len(pending), nv.C.byref(out), st))
                if not os.environ.get("LIDBOX_GEMM_NO_CARRY"):
                    assert nv.lib.lidbox_gemm_last_carried() == len(pending) + (0 if li < 2 else 1)
                pending = [out] if out.nblocks else []
                assert bool(out.nblocks) == (li < 2)          # the M = 256 pair launches hand their reduce on, the conv layer does not
            else:
                nv.check(nv.lib.lidbox_gemm_nt_tn(*args, st))
            outs += [dx, dw, db]
        assert not pending
        torch.cuda.synchronize()
        return outs

    a, b = run(True), run(False)
    for u, v in zip(a, b):
        assert torch.equal(u, v)
    monkeypatch.setenv("LIDBOX_GEMM_NO_CARRY", "1")
    for u, v in zip(run(True), b):
        assert torch.equal(u, v)


def test_zero_fill_job_alone_and_carried():
    """lidbox_zero_job: `batch` runs of zeros as a job without slices; run on its own and in the leading workgroups of a GEMM
    launch it clears exactly its runs (the rows a strided Conv1D's accumulating dgrad groups do not all cover)"""
    from lidbox_amd import _native as nv
    st = nv.current_stream()
    B, Tp, cin, rows = 37, 20, 64, 2
    for carried in (False, True):
        buf = torch.full((B, Tp, cin), 7.0, device="cuda")
        job = nv.ReduceJob()
        nv.check(nv.lib.lidbox_zero_job(nv.C.c_void_p(buf.data_ptr() + 4 * (Tp - rows) * cin), Tp * cin, rows * cin, B, nv.C.byref(job)))
        if carried:
            rng = np.random.default_rng(2)
            M, K, N = 1024, 256, 256
            a, w = _dev(rng.standard_normal((M, K))), _dev(rng.standard_normal((N, K)))
            c = torch.zeros((M, N), device="cuda")
            ws = _garbage_ws(max(nv.lib.lidbox_gemm_rows_workspace(M, N, K), 16) + 1024)
            nv.check(nv.lib.lidbox_gemm_nt_carry(_rows(a, 0, K, 1, M), nv.ptr(w), K, _rows(c, 0, N, 1, M), K, N, nv.EPI_NONE, None, nv.ptr(ws),
                                                 ws.numel(), nv.C.byref(job), 1, st))
            assert nv.lib.lidbox_gemm_last_carried() == 1
            _close(c.cpu().numpy(), a.cpu().numpy().astype(np.float64) @ w.cpu().numpy().astype(np.float64).T)
        else:
            nv.check(nv.lib.lidbox_reduce_jobs_run(nv.C.byref(job), 1, st))
        out = buf.cpu().numpy()
        assert (out[:, Tp - rows:, :] == 0).all() and (out[:, :Tp - rows, :] == 7.0).all()
    rc = nv.lib.lidbox_zero_job(nv.C.c_void_p(buf.data_ptr() + 4), Tp * cin, rows * cin, B, nv.C.byref(job))
    assert rc != 0 and b"aligned" in nv.lib.lidbox_hip_last_error()


@pytest.mark.parametrize("bn", ["64", "128"])
@pytest.mark.parametrize("M,K,N,epi", [(8448, 512, 512, "mask"), (8448, 1500, 512, "none"), (4100, 516, 200, "accmask"), (33 * 130, 512, 1536, "mask")])
def test_eight_wave_dgrad_tiles_match_float64(M, K, N, epi, bn, monkeypatch):
    """gemm_rows_dma8_kernel (128-row tiles worked by eight waves: gemm_dma8.h) through lidbox_gemm_nt / lidbox_gemm_nt_carry:
    ragged row counts, K tails (1500, 516), N that is no multiple of the tile, mask / accumulate epilogues, the streamed
    remainder and a carried reduce; equal to float64 at round-off, run-to-run bit-identical, and the carried job's result equal
    to the plain reduce.  Reference: the dgrad of Conv1D, lidbox/models/xvector.py:38-43,53-57."""
    from lidbox_amd import _native as nv
    monkeypatch.setenv("LIDBOX_GEMM_NT8", bn)
    rng = np.random.default_rng(M + K)
    A, W = rng.standard_normal((M, K)), rng.standard_normal((N, K)) * 0.1
    mask, old = rng.standard_normal((M, N)), rng.standard_normal((M, N)).astype(np.float32)
    a, w, mk = _dev(A), _dev(W), _dev(mask)
    st = nv.current_stream()
    ws = _garbage_ws(max(nv.lib.lidbox_gemm_rows_workspace(M, N, K), 16) + 1024)
    e = {"mask": nv.EPI_RELU_MASK, "none": nv.EPI_NONE, "accmask": nv.EPI_ACCUM_RELU_MASK}[epi]
    ref = A @ W.T
    if epi != "none":
        ref = ref * (mask > 0)
    if epi == "accmask":
        ref = ref + old.astype(np.float64)
    # a pending wgrad-style job to carry: 3 slices of a [64, 128] matrix
    P = _dev(rng.standard_normal((3, 64 * 128)))
    outs = []
    for rep in range(2):
        c = _dev(old).clone() if epi == "accmask" else torch.full((M, N), 9.0, device="cuda")
        dw = torch.full((64, 128), 5.0, device="cuda")
        job = nv.ReduceJob(P.data_ptr(), None, dw.data_ptr(), None, 64 * 128, 128, 3, 128, 0, 8)
        nv.check(nv.lib.lidbox_gemm_nt_carry(_rows(a, 0, K, 1, M), nv.ptr(w), K, _rows(c, 0, N, 1, M), K, N, e, nv.ptr(mk) if epi != "none" else None,
                                             nv.ptr(ws), ws.numel(), nv.C.byref(job), 1, st))
        assert nv.lib.lidbox_gemm_last_family() == 3 and nv.lib.lidbox_gemm_last_carried() == 1
        torch.cuda.synchronize()
        outs.append((c, dw))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    _close(outs[0][0].cpu().numpy(), ref)
    Pn = P.cpu().numpy().astype(np.float32)
    assert np.array_equal(outs[0][1].cpu().numpy().ravel(), (Pn[0] + Pn[1]) + Pn[2])


@pytest.mark.parametrize("bn", ["64", "128"])
@pytest.mark.parametrize("M,K,N", [(50688 // 8, 200, 512), (8448, 512, 1500), (4100, 516, 200)])
def test_eight_wave_forward_tiles_match_float64(M, K, N, bn, monkeypatch):
    """the nn form of gemm_rows_dma8_kernel (B [K][N] K-outer, bias + ReLU epilogue): K tails (200, 516), N = 1500 / 200 (partial
    column tiles), ragged rows; float64 at round-off and bit-identical from run to run.  Built and measured in round 4, not
    adopted for the forward launches (profiles/r04_nt8_ab.txt); reachable through a tuned-table entry with waves = 8."""
    from lidbox_amd import _native as nv
    monkeypatch.setenv("LIDBOX_GEMM_NN8", bn)
    rng = np.random.default_rng(M + N)
    A, W, b = rng.standard_normal((M, K)), rng.standard_normal((K, N)) * 0.1, rng.standard_normal(N)
    a, w, bias = _dev(A), _dev(W), _dev(b)
    st = nv.current_stream()
    ws = _garbage_ws(max(nv.lib.lidbox_gemm_rows_workspace(M, N, K), 16) + 1024)
    outs = []
    for rep in range(2):
        c = torch.full((M, N), 9.0, device="cuda")
        nv.check(nv.lib.lidbox_gemm_nn(_rows(a, 0, K, 1, M), nv.ptr(w), N, _rows(c, 0, N, 1, M), K, N, nv.EPI_BIAS_RELU, nv.ptr(bias), nv.ptr(ws),
                                       ws.numel(), st))
        assert nv.lib.lidbox_gemm_last_family() == 3
        outs.append(c)
    assert torch.equal(outs[0], outs[1])
    _close(outs[0].cpu().numpy(), np.maximum(A @ W + b, 0.0))
