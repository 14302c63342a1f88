"""
GPU parity of the bf16-compute GEMM family (lidbox_gemm_bf16_nn/_nt/_tn, BASELINE config 5).

Contract under test (include/lidbox_hip.h): operands are rounded to bfloat16 round-to-nearest-even,
products accumulate in fp32, epilogues / bias gradients are fp32.  A product of two bf16 values is
exact in fp32, so the result must equal the float64 GEMM of the bf16-ROUNDED operands to fp32
summation round-off -- rel 2e-5 of the result scale, the same tolerance as the fp32 family -- which
is far tighter than the ~4e-3 relative step of bf16 itself: a wrong rounding mode, a dropped K tail
or a transposed fragment cannot pass.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev(x, dtype=np.float32):
    return torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype))).cuda()


def _rows(t, bs, rs, batch, rpb, off_floats=0):
    from lidbox_amd import _native as nv
    return nv.Rows(t.data_ptr() + 4 * off_floats, bs, rs, batch, rpb)


def _bf16(x):
    """float64 value of x after fp32 -> bf16 RNE rounding"""
    return torch.from_numpy(np.asarray(x, np.float32)).bfloat16().double().numpy()


def _close(got, ref, rel=2e-5):
    scale = max(1e-30, float(np.abs(ref).max()))
    err = float(np.abs(got - ref).max())
    assert err <= rel * scale, (err, scale)


def _ws(nbytes):
    return torch.empty(max(16, nbytes), dtype=torch.uint8, device="cuda")


@pytest.mark.parametrize("M,K,N", [(128, 32, 128), (200, 200, 512), (33, 1536, 512), (1000, 260, 40), (5, 4, 8),
                                   (256, 3000, 512), (130, 64, 1500), (256, 512, 4), (512, 512, 100), (1, 4, 4)])
def test_bf16_gemm_nn(M, K, N):
    from lidbox_amd import _native as nv
    rng = np.random.default_rng(M * 7 + K)
    A, B, bias = rng.standard_normal((M, K)), rng.standard_normal((K, N)), rng.standard_normal(N)
    ref = _bf16(A) @ _bf16(B)
    a, b, bi = _dev(A), _dev(B), _dev(bias)
    c = torch.full((M, N), 7.0, device="cuda")
    st = nv.current_stream()
    ra, rc = _rows(a, 0, K, 1, M), _rows(c, 0, N, 1, M)
    nv.check(nv.lib.lidbox_gemm_bf16_nn(ra, nv.ptr(b), N, rc, K, N, nv.EPI_NONE, None, None, 0, st))
    _close(c.cpu().numpy(), ref)
    # it really is a bf16 product: the fp32 GEMM of the unrounded operands differs visibly once K is non-trivial
    if K >= 200:
        assert np.abs(c.cpu().numpy() - A.astype(np.float32) @ B.astype(np.float32)).max() > 1e-3
    nv.check(nv.lib.lidbox_gemm_bf16_nn(ra, nv.ptr(b), N, rc, K, N, nv.EPI_BIAS_RELU, nv.ptr(bi), None, 0, st))
    _close(c.cpu().numpy(), np.maximum(ref + np.float32(bias), 0))
    # split-K workspace path (small-M problems); epilogue fused into the reduce
    wsb = nv.lib.lidbox_gemm_bf16_rows_workspace(M, N, K)
    ws = _ws(wsb)
    c.fill_(-3.0)
    nv.check(nv.lib.lidbox_gemm_bf16_nn(ra, nv.ptr(b), N, rc, K, N, nv.EPI_BIAS, nv.ptr(bi), nv.ptr(ws), ws.numel(), st))
    _close(c.cpu().numpy(), ref + np.float32(bias))
    c2 = torch.empty_like(c)
    nv.check(nv.lib.lidbox_gemm_bf16_nn(ra, nv.ptr(b), N, _rows(c2, 0, N, 1, M), K, N, nv.EPI_BIAS, nv.ptr(bi), nv.ptr(ws),
                                        ws.numel(), st))
    assert torch.equal(c, c2)                              # deterministic


@pytest.mark.parametrize("M,K,N", [(128, 512, 1024), (99, 512, 1536), (7, 8, 4), (256, 4, 512), (300, 100, 3000),
                                   (256, 512, 512)])
def test_bf16_gemm_nt_and_epilogues(M, K, N):
    from lidbox_amd import _native as nv
    rng = np.random.default_rng(M + 3 * N)
    A, B = rng.standard_normal((M, K)), rng.standard_normal((N, K))
    mask = rng.standard_normal((M, N))
    ref = _bf16(A) @ _bf16(B).T
    a, b, mk = _dev(A), _dev(B), _dev(mask)
    c = torch.full((M, N), 2.0, device="cuda")
    st = nv.current_stream()
    ra, rc = _rows(a, 0, K, 1, M), _rows(c, 0, N, 1, M)
    wsb = nv.lib.lidbox_gemm_bf16_rows_workspace(M, N, K)
    ws = _ws(wsb)
    nv.check(nv.lib.lidbox_gemm_bf16_nt(ra, nv.ptr(b), K, rc, K, N, nv.EPI_NONE, None, None, 0, st))
    _close(c.cpu().numpy(), ref)
    nv.check(nv.lib.lidbox_gemm_bf16_nt(ra, nv.ptr(b), K, rc, K, N, nv.EPI_RELU_MASK, nv.ptr(mk), nv.ptr(ws), ws.numel(), st))
    _close(c.cpu().numpy(), ref * (mask > 0))
    c.fill_(-2.0)
    nv.check(nv.lib.lidbox_gemm_bf16_nt(ra, nv.ptr(b), K, rc, K, N, nv.EPI_ACCUM, None, nv.ptr(ws), ws.numel(), st))
    _close(c.cpu().numpy(), ref - 2.0)
    c.fill_(-2.0)
    nv.check(nv.lib.lidbox_gemm_bf16_nt(ra, nv.ptr(b), K, rc, K, N, nv.EPI_ACCUM_RELU_MASK, nv.ptr(mk), None, 0, st))
    _close(c.cpu().numpy(), ref * (mask > 0) - 2.0)


@pytest.mark.parametrize("M,K1,N", [(4096, 200, 512), (1000, 1536, 512), (256, 3000, 512), (50, 8, 4), (8448, 512, 1500),
                                    (37, 132, 260)])
def test_bf16_gemm_tn_and_bias_grad(M, K1, N):
    from lidbox_amd import _native as nv
    rng = np.random.default_rng(M + N)
    A, B = rng.standard_normal((M, K1)), rng.standard_normal((M, N))
    ref = _bf16(A).T @ _bf16(B)
    a, b = _dev(A), _dev(B)
    c = torch.full((K1, N), 3.0, device="cuda")
    st = nv.current_stream()
    wsb = nv.lib.lidbox_gemm_bf16_tn_workspace(M, K1, N)
    ws = _ws(wsb)
    ra, rb = _rows(a, 0, K1, 1, M), _rows(b, 0, N, 1, M)
    nv.check(nv.lib.lidbox_gemm_bf16_tn(ra, rb, nv.ptr(c), N, K1, N, 0, None, nv.ptr(ws), wsb, st))
    _close(c.cpu().numpy(), ref)
    nv.check(nv.lib.lidbox_gemm_bf16_tn(ra, rb, nv.ptr(c), N, K1, N, 1, None, nv.ptr(ws), wsb, st))
    _close(c.cpu().numpy(), 2 * ref)
    c2, c3 = torch.empty_like(c), torch.empty_like(c)
    nv.check(nv.lib.lidbox_gemm_bf16_tn(ra, rb, nv.ptr(c2), N, K1, N, 0, None, nv.ptr(ws), wsb, st))
    bg = torch.full((N,), 9.0, device="cuda")
    nv.check(nv.lib.lidbox_gemm_bf16_tn(ra, rb, nv.ptr(c3), N, K1, N, 0, nv.ptr(bg), nv.ptr(ws), wsb, st))
    assert torch.equal(c2, c3)                              # deterministic, with or without the fused bias gradient
    # the bias gradient is summed in fp32 from the UNROUNDED operand
    _close(bg.cpu().numpy(), np.asarray(B, np.float32).astype(np.float64).sum(axis=0), 1e-5)


def test_bf16_gemm_batched_implicit_rows_match_fp32_layout():
    """implicit causal rows (batch > 1, overlapping windows, strided) -- same addressing as the fp32 family"""
    from lidbox_amd import _native as nv
    rng = np.random.default_rng(5)
    Bn, T, C, k, s, Co = 3, 50, 8, 3, 2, 12
    Tp = T + k - 1
    x = np.zeros((Bn, Tp, C))
    x[:, k - 1:] = rng.standard_normal((Bn, T, C))
    W = rng.standard_normal((k * C, Co))
    To = (T - 1) // s + 1
    cols = np.stack([x[:, t * s:t * s + k].reshape(Bn, k * C) for t in range(To)], axis=1)      # [B, To, k*C]
    ref = _bf16(cols) @ _bf16(W)
    xd, wd = _dev(x), _dev(W)
    y = torch.zeros(Bn, To, Co, device="cuda")
    st = nv.current_stream()
    ra = _rows(xd, Tp * C, s * C, Bn, To)
    ry = _rows(y, To * Co, Co, Bn, To)
    nv.check(nv.lib.lidbox_gemm_bf16_nn(ra, nv.ptr(wd), Co, ry, k * C, Co, nv.EPI_NONE, None, None, 0, st))
    _close(y.cpu().numpy(), ref)
    # wgrad over the same implicit rows: dW = cols^T . dY, with fused bias gradient
    dY = rng.standard_normal((Bn, To, Co))
    dyd = _dev(dY)
    dW = torch.zeros(k * C, Co, device="cuda")
    db = torch.zeros(Co, device="cuda")
    wsb = nv.lib.lidbox_gemm_bf16_tn_workspace(Bn * To, k * C, Co)
    ws = _ws(wsb)
    nv.check(nv.lib.lidbox_gemm_bf16_tn(ra, _rows(dyd, To * Co, Co, Bn, To), nv.ptr(dW), Co, k * C, Co, 0, nv.ptr(db),
                                        nv.ptr(ws), wsb, st))
    _close(dW.cpu().numpy(), _bf16(cols).reshape(-1, k * C).T @ _bf16(dY).reshape(-1, Co))
    _close(db.cpu().numpy(), np.asarray(dY, np.float32).astype(np.float64).reshape(-1, Co).sum(axis=0), 1e-5)
    # rows_per_batch == 1 (every row crosses an utterance boundary)
    ra1 = _rows(xd, Tp * C, s * C, Bn, 1)
    y1 = torch.zeros(Bn, 1, Co, device="cuda")
    nv.check(nv.lib.lidbox_gemm_bf16_nn(ra1, nv.ptr(wd), Co, _rows(y1, Co, Co, Bn, 1), k * C, Co, nv.EPI_NONE, None, None, 0, st))
    _close(y1.cpu().numpy()[:, 0], ref[:, 0])
    dW1 = torch.zeros(k * C, Co, device="cuda")
    wsb1 = nv.lib.lidbox_gemm_bf16_tn_workspace(Bn, k * C, Co)
    ws1 = _ws(wsb1)
    nv.check(nv.lib.lidbox_gemm_bf16_tn(ra1, _rows(dyd, To * Co, Co, Bn, 1), nv.ptr(dW1), Co, k * C, Co, 0, None, nv.ptr(ws1),
                                        wsb1, st))
    _close(dW1.cpu().numpy(), _bf16(cols[:, 0]).T @ _bf16(dY[:, 0]))


def test_bf16_gemm_rejects_unaligned_shapes():
    from lidbox_amd import _native as nv
    a = torch.zeros(8, 7, device="cuda")
    b = torch.zeros(7, 8, device="cuda")
    c = torch.zeros(8, 8, device="cuda")
    rc = nv.lib.lidbox_gemm_bf16_nn(_rows(a, 0, 7, 1, 8), nv.ptr(b), 8, _rows(c, 0, 8, 1, 8), 7, 8, nv.EPI_NONE, None, None, 0,
                                    nv.current_stream())
    assert rc != 0 and "multiples of 4" in nv.last_error()


@pytest.mark.parametrize("M,K,N", [(17024, 1536, 512), (17000, 1024, 512), (8500, 1280, 1024), (17024, 512, 512)])
def test_bf16_gemm_tail_split_matches_single_launch(M, K, N):
    """More tiles than one round of workgroup slots with a few left over: the launch is cut into a whole-round prefix and a
    K-split remainder (gemm_bf16.hip: launch_rows16).  Same results as the product of the rounded operands, for the
    forward epilogue and for the backward one (mask + accumulate) over rows with a gap between utterances."""
    from lidbox_amd import _native as nv
    rng = np.random.default_rng(M + N)
    A, B, Bt = rng.standard_normal((M, K)), rng.standard_normal((K, N)), rng.standard_normal((N, K))
    bias, mask = rng.standard_normal(N), rng.standard_normal((M, N))
    a, b, bt, bi, mk = _dev(A), _dev(B), _dev(Bt), _dev(bias), _dev(mask)
    st = nv.current_stream()
    ws = _ws(256 << 20)
    c = torch.zeros((M, N), device="cuda")
    nv.check(nv.lib.lidbox_gemm_bf16_nn(_rows(a, 0, K, 1, M), nv.ptr(b), N, _rows(c, 0, N, 1, M), K, N, nv.EPI_BIAS_RELU,
                                        nv.ptr(bi), nv.ptr(ws), ws.numel(), st))
    _close(c.cpu().numpy(), np.maximum(_bf16(A) @ _bf16(B) + np.float32(bias), 0))
    # utterances of 100 rows in a [B, 103, N] buffer (3 untouched rows in front of each)
    Bn, R, Rp = M // 100, 100, 103
    cb = torch.full((Bn, Rp, N), 7.0, device="cuda")
    mkb = torch.zeros((Bn, Rp, N), device="cuda")
    mkb[:, 3:, :] = mk[:Bn * R].reshape(Bn, R, N)
    Cd = nv.Rows(cb[:, 3:, :].data_ptr(), Rp * N, N, Bn, R)
    nv.check(nv.lib.lidbox_gemm_bf16_nt(_rows(a, 0, K, 1, Bn * R), nv.ptr(bt), K, Cd, K, N, nv.EPI_ACCUM_RELU_MASK,
                                        nv.C.c_void_p(mkb[:, 3:, :].data_ptr()), nv.ptr(ws), ws.numel(), st))
    got = cb.cpu().numpy()
    _close(got[:, 3:, :].reshape(Bn * R, N), (_bf16(A[:Bn * R]) @ _bf16(Bt).T) * (mask[:Bn * R] > 0) + 7.0)
    assert np.all(got[:, :3, :] == 7.0)


@pytest.mark.parametrize("variant", [None, "64,64,2", "64,64,3", "128,64,2", "64,128,2", "128,128,2", "128,128,3",
                                     "256,256,2", "256,256,1", "256,128,2", "256,128,1"])
@pytest.mark.parametrize("M,K,N", [(128, 32, 128), (200, 200, 512), (33, 1536, 512), (1000, 264, 40), (5, 8, 8),
                                   (256, 3000, 512), (700, 512, 1024), (1, 8, 4), (50688 // 8, 200, 512), (300, 72, 1500), (257, 64, 36)])
def test_bf16_storage_gemm_nt_and_shadow_output(M, K, N, variant, monkeypatch):
    """lidbox_gemm_bf16s_nt: operands already bf16 in HBM ([M][K] and [N][K]); same numbers as the fp32-source kernel on the
    unrounded originals; the bf16 shadow of C equals bf16(C); split-K, epilogues, converters.
    variant: None = the library's own choice (small problems: the register-staged 128 x 128 kernel), "bm,bn,stages" = that
    instantiation of the LDS-DMA kernel (csrc/gemm16_dma.h; K tails of 8 .. 56 past a 64-deep step, ragged M / N edges);
    "256,bn,sub" = the eight-wave ping-pong tile (csrc/gemm16_pp.h)"""
    from lidbox_amd import _native as nv
    if variant is not None:
        monkeypatch.setenv("LIDBOX_GEMM16S_DMA", variant)
    rng = np.random.default_rng(M * 5 + K)
    A, B, bias = rng.standard_normal((M, K)), rng.standard_normal((N, K)), rng.standard_normal(N)
    ref = _bf16(A) @ _bf16(B).T
    st = nv.current_stream()
    a32, b32, bi = _dev(A), _dev(B), _dev(bias)
    a16 = torch.empty((M, K), dtype=torch.bfloat16, device="cuda")
    b16 = torch.empty((N, K), dtype=torch.bfloat16, device="cuda")
    nv.check(nv.lib.lidbox_f32_to_bf16(nv.ptr(a32), nv.ptr(a16), M * K, st))
    assert torch.equal(a16, a32.bfloat16())
    # B through the transposing converter: from the [K][N] layout a forward weight has
    bT = _dev(B.T)
    nv.check(nv.lib.lidbox_transpose_f32_to_bf16(nv.ptr(bT), K, N, N, nv.ptr(b16), K, st))
    assert torch.equal(b16, b32.bfloat16())
    c = torch.full((M, N), 7.0, device="cuda")
    c16 = torch.full((M, N), 3.0, dtype=torch.bfloat16, device="cuda")
    ra = nv.Rows(a16.data_ptr(), 0, K, 1, M)
    rc = _rows(c, 0, N, 1, M)
    nv.check(nv.lib.lidbox_gemm_bf16s_nt(ra, nv.ptr(b16), K, rc, nv.ptr(c16), K, N, nv.EPI_NONE, None, None, 0, st))
    _close(c.cpu().numpy(), ref)
    assert torch.equal(c16, c.bfloat16())
    out3 = (nv.C.c_int * 3)()
    nv.check(nv.lib.lidbox_gemm_bf16s_last_variant(out3))
    if variant:
        assert list(out3) == [int(v) for v in variant.split(",")]
    # the fp32-source kernel on the same data gives the same product (to summation order)
    c_old = torch.zeros((M, N), device="cuda")
    if K % 4 == 0:
        nv.check(nv.lib.lidbox_gemm_bf16_nt(_rows(a32, 0, K, 1, M), nv.ptr(b32), K, _rows(c_old, 0, N, 1, M), K, N, nv.EPI_NONE,
                                            None, None, 0, st))
        _close(c.cpu().numpy(), c_old.cpu().double().numpy(), rel=1e-5)
    # epilogues + split-K workspace; no shadow requested
    wsb = max(16, nv.lib.lidbox_gemm_bf16_rows_workspace(M, N, K))
    ws = _ws(wsb)
    nv.check(nv.lib.lidbox_gemm_bf16s_nt(ra, nv.ptr(b16), K, rc, None, K, N, nv.EPI_BIAS_RELU, nv.ptr(bi), nv.ptr(ws), wsb, st))
    _close(c.cpu().numpy(), np.maximum(ref + np.float32(bias), 0))
    mask = _dev(rng.standard_normal((M, N)))
    c.fill_(1.0)
    nv.check(nv.lib.lidbox_gemm_bf16s_nt(ra, nv.ptr(b16), K, rc, nv.ptr(c16), K, N, nv.EPI_ACCUM_RELU_MASK, nv.ptr(mask), nv.ptr(ws), wsb, st))
    want = 1.0 + ref * (mask.cpu().numpy() > 0)
    _close(c.cpu().numpy(), want)
    assert torch.equal(c16, c.bfloat16())


@pytest.mark.parametrize("variant", ["256,256,2", "256,128,2", "256,256,2,3"])
def test_bf16_pingpong_tile_carries_reduce_jobs_and_splits_k(variant, monkeypatch):
    """the eight-wave ping-pong tile (csrc/gemm16_pp.h, 512-thread workgroups) as the carrier of a pending wgrad slice sum
    (lidbox_gemm_bf16s_tn_partial + lidbox_gemm_bf16s_nt_carry: the leading workgroups' first 256 threads run the job) and along
    a K split ("bm,bn,sub,splits": grid.y > 1, partial sums through the workspace + rows_reduce_kernel): the same bits as the
    separate launches / the unsplit launch of the same variant up to fp32 summation order, the float64 product to round-off;
    ReLU mask from bf16 data, shadow-only output, batched rows that cross utterance boundaries inside a 32-row strip"""
    from lidbox_amd import _native as nv
    monkeypatch.setenv("LIDBOX_GEMM16S_DMA", variant)
    rng = np.random.default_rng(17)
    Bn, R, Co, N, K1 = 9, 61, 264, 520, 200                       # M = 549 rows in 9 utterances of 61 (+ 3 pad rows each)
    M = Bn * R
    dy16 = _dev(rng.standard_normal((M, Co))).bfloat16()
    w16 = (_dev(rng.standard_normal((N, Co))) * 0.1).bfloat16()
    x16 = _dev(rng.standard_normal((M, K1))).bfloat16()
    mask16 = _dev(np.maximum(rng.standard_normal((Bn, R + 3, N)), 0)).bfloat16()
    st = nv.current_stream()
    wsb = max(16, nv.lib.lidbox_gemm_bf16_rows_workspace(M, N, Co)) + (3 * M * N * 4 if variant.count(",") == 3 else 0)
    ws1 = _ws(wsb)
    ws2 = _ws(max(16, nv.lib.lidbox_gemm_bf16s_tn_workspace(M, K1, Co)))
    ra, rb = nv.Rows(x16.data_ptr(), 0, K1, 1, M), nv.Rows(dy16.data_ptr(), 0, Co, 1, M)

    def run(carry):
        dx16 = torch.full((Bn, R + 3, N), 5.0, dtype=torch.bfloat16, device="cuda")
        dw = torch.full((K1, Co), 3.0, device="cuda")
        db = torch.full((Co,), 3.0, device="cuda")
        Cd = nv.Rows(None, (R + 3) * N, N, Bn, R)                   # shadow only, rows behind 3 pad rows per utterance
        c16 = nv.C.c_void_p(dx16.data_ptr() + 2 * 3 * N)
        mk = nv.C.c_void_p(mask16.data_ptr() + 2 * 3 * N)
        epi = nv.EPI_RELU_MASK | nv.EPI_MASK_BF16
        job = nv.ReduceJob()
        nv.check(nv.lib.lidbox_gemm_bf16s_tn_partial(ra, rb, nv.ptr(dw), Co, K1, Co, 0, nv.ptr(db), nv.ptr(ws2), ws2.numel(), nv.C.byref(job), st))
        if carry:
            nv.check(nv.lib.lidbox_gemm_bf16s_nt_carry(rb, nv.ptr(w16), Co, Cd, c16, Co, N, epi, mk, nv.ptr(ws1), ws1.numel(), nv.C.byref(job), 1, st))
            assert nv.lib.lidbox_gemm_bf16s_last_carried() == 1
        else:
            nv.check(nv.lib.lidbox_reduce_jobs_run(nv.C.byref(job), 1, st))
            nv.check(nv.lib.lidbox_gemm_bf16s_nt(rb, nv.ptr(w16), Co, Cd, c16, Co, N, epi, mk, nv.ptr(ws1), ws1.numel(), st))
        out3 = (nv.C.c_int * 3)()
        nv.check(nv.lib.lidbox_gemm_bf16s_last_variant(out3))
        assert list(out3) == [int(v) for v in variant.split(",")[:3]]
        torch.cuda.synchronize()
        return dx16, dw, db
    a, b = run(True), run(False)
    for u, v in zip(a, b):
        assert torch.equal(u, v)
    dx16, dw, db = a
    assert bool((dx16[:, :3, :] == 5.0).all())                     # pad rows untouched
    ref = (dy16.double() @ w16.double().T).reshape(Bn, R, N) * (mask16[:, 3:, :] > 0)
    assert torch.equal(dx16[:, 3:, :], ref.float().bfloat16()) or float((dx16[:, 3:, :].double() - ref).abs().max()) <= 2.0 ** -7 * float(ref.abs().max())
    refw = x16.double().T @ dy16.double()
    assert float((dw.double() - refw).abs().max()) <= 2e-5 * float(refw.abs().max())
    assert float((db.double() - dy16.double().sum(0)).abs().max()) <= 2e-5 * float(dy16.double().sum(0).abs().max())


@pytest.mark.parametrize("forced", [True, False])
def test_bf16_two_problems_in_one_grid_equal_two_launches(forced, monkeypatch):
    """lidbox_gemm_bf16s_nt_pair_carry: the two row residues of a stride-2 convolution's output-stationary dgrad (windows of Q = 2 / 1
    output-gradient rows against the stacked taps, interleaved output rows, ReLU mask from bf16 data, shadow-only output) as ONE grid of
    the ping-pong tile == the two launches one after the other, bit for bit; the carried wgrad reduce runs in the same launch; a pair
    the policy keeps off the tile falls back to the two calls.  forced: small ragged shapes on the forced tile;
    otherwise frame2's own shapes at 256 utterances, where the policy picks the tile by itself."""
    from lidbox_amd import _native as nv
    if forced:
        monkeypatch.setenv("LIDBOX_GEMM16S_DMA", "256,256,2,1")           # the tile, no K split (small problems get one otherwise)
        Bn, To, Co, cin = 7, 45, 264, 520
    else:
        Bn, To, Co, cin = 256, 99, 512, 512
    rng = np.random.default_rng(23)
    s_, pad = 2, 2
    Tin = To * s_
    # dY shadow with one zero row ahead of every utterance (the window of residue 0 starts one row early) and one behind
    dy16 = torch.zeros((Bn, To + 2, Co), dtype=torch.bfloat16, device="cuda")
    dy16[:, 1:To + 1] = _dev(rng.standard_normal((Bn, To, Co))).bfloat16()
    w_even = (_dev(rng.standard_normal((cin, 2 * Co))) * 0.05).bfloat16()       # taps {2, 0} side by side
    w_odd = (_dev(rng.standard_normal((cin, Co))) * 0.05).bfloat16()            # tap {1}
    act16 = _dev(rng.standard_normal((Bn, pad + Tin, cin))).bfloat16()
    K1 = 200
    x16 = _dev(rng.standard_normal((Bn * To, K1))).bfloat16()
    st = nv.current_stream()
    M = Bn * To
    wsb = max(16, nv.lib.lidbox_gemm_bf16_rows_workspace(M, cin, 2 * Co))
    ws1 = _ws(wsb)
    ws2 = _ws(max(16, nv.lib.lidbox_gemm_bf16s_tn_workspace(M, K1, Co)))
    epi = nv.EPI_RELU_MASK | nv.EPI_MASK_BF16

    def problems(dx16):
        out = []
        for rho, Q, w in ((0, 2, w_even), (1, 1, w_odd)):
            A = nv.Rows(dy16.data_ptr() + 2 * (1 - (Q - 1)) * Co, (To + 2) * Co, Co, Bn, To)
            p0 = pad + rho
            Cd = nv.Rows(None, (pad + Tin) * cin, s_ * cin, Bn, To)
            out.append((A, nv.ptr(w), Q * Co, Cd, nv.C.c_void_p(dx16.data_ptr() + 2 * p0 * cin), Q * Co, cin, epi,
                        nv.C.c_void_p(act16.data_ptr() + 2 * p0 * cin)))
        return out

    def run(pair):
        dx16 = torch.full((Bn, pad + Tin, cin), 5.0, dtype=torch.bfloat16, device="cuda")
        dw, db = torch.full((K1, Co), 3.0, device="cuda"), torch.full((Co,), 3.0, device="cuda")
        job = nv.ReduceJob()
        ra = nv.Rows(x16.data_ptr(), 0, K1, 1, M)
        rb = nv.Rows(dy16.data_ptr() + 2 * Co, (To + 2) * Co, Co, Bn, To)
        nv.check(nv.lib.lidbox_gemm_bf16s_tn_partial(ra, rb, nv.ptr(dw), Co, K1, Co, 0, nv.ptr(db), nv.ptr(ws2), ws2.numel(), nv.C.byref(job), st))
        c0, c1 = problems(dx16)
        if pair:
            nv.check(nv.lib.lidbox_gemm_bf16s_nt_pair_carry(*c0, *c1, nv.ptr(ws1), ws1.numel(), nv.C.byref(job), 1, st))
            took = nv.lib.lidbox_gemm_bf16s_last_pair()
            assert nv.lib.lidbox_gemm_bf16s_last_carried() == 1
        else:
            nv.check(nv.lib.lidbox_gemm_bf16s_nt_carry(*c0, nv.ptr(ws1), ws1.numel(), nv.C.byref(job), 1, st))
            nv.check(nv.lib.lidbox_gemm_bf16s_nt(*c1, nv.ptr(ws1), ws1.numel(), st))
            took = 0
        torch.cuda.synchronize()
        return dx16, dw, db, took
    a, b = run(True), run(False)
    assert a[3] == 1
    for u, v in zip(a[:3], b[:3]):
        assert torch.equal(u, v)
    dx16 = a[0]
    assert bool((dx16[:, :pad] == 5.0).all())                            # the causal pad rows are nobody's output
    # against the float64 product of the stored values: even rows = dY[u-1] W2^T + dY[u] W0^T (stacked), odd rows = dY[u] W1^T
    dyd = dy16.double()
    win = torch.cat([dyd[:, 0:To], dyd[:, 1:To + 1]], dim=2)             # rows u-1 | u
    even = (win @ w_even.double().T) * (act16[:, pad::2].double() > 0)
    odd = (dyd[:, 1:To + 1] @ w_odd.double().T) * (act16[:, pad + 1::2].double() > 0)
    for got, ref in ((dx16[:, pad::2], even), (dx16[:, pad + 1::2], odd)):
        assert float((got.double() - ref).abs().max()) <= 2.0 ** -7 * float(ref.abs().max())
    if forced:
        monkeypatch.delenv("LIDBOX_GEMM16S_DMA")                         # the policy's own choice at this size: four-wave tiles
        c, d = run(True), run(False)
        assert c[3] == 0                                                 # fell back to two launches
        for u, v in zip(c[:3], d[:3]):
            assert torch.equal(u, v)


def test_bf16_storage_gemm_implicit_rows_and_errors():
    """strided causal windows over a bf16 shadow [B, pad + T, C] (Conv1D k = 3, stride 2) and the alignment rules"""
    from lidbox_amd import _native as nv
    rng = np.random.default_rng(77)
    Bn, T, C, k, s_, Co = 5, 21, 16, 3, 2, 24
    pad = k - 1
    x = np.zeros((Bn, pad + T, C))
    x[:, pad:] = rng.standard_normal((Bn, T, C))
    W = rng.standard_normal((k * C, Co))
    To = (T - 1) // s_ + 1
    idx = np.arange(To)[:, None] * s_ + np.arange(k)[None, :]
    col = _bf16(x)[:, idx, :].reshape(Bn, To, k * C)
    ref = col @ _bf16(W)
    st = nv.current_stream()
    x16 = _dev(x).bfloat16()
    wT16 = _dev(W.T).bfloat16().contiguous()                 # [Co][k*C]
    out = torch.zeros((Bn, To, Co), device="cuda")
    out16 = torch.zeros((Bn, To, Co), dtype=torch.bfloat16, device="cuda")
    ra = nv.Rows(x16.data_ptr(), (pad + T) * C, s_ * C, Bn, To)
    nv.check(nv.lib.lidbox_gemm_bf16s_nt(ra, nv.ptr(wT16), k * C, _rows(out, To * Co, Co, Bn, To), nv.ptr(out16), k * C, Co,
                                         nv.EPI_NONE, None, None, 0, st))
    _close(out.cpu().numpy(), ref)
    assert torch.equal(out16, out.bfloat16())
    with pytest.raises(ValueError):                          # K not a multiple of 8
        nv.check(nv.lib.lidbox_gemm_bf16s_nt(ra, nv.ptr(wT16), k * C, _rows(out, To * Co, Co, Bn, To), None, k * C - 4, Co,
                                             nv.EPI_NONE, None, None, 0, st))
    with pytest.raises(ValueError):                          # row stride not a multiple of 8 elements
        nv.check(nv.lib.lidbox_gemm_bf16s_nt(nv.Rows(x16.data_ptr(), (pad + T) * C, 12, Bn, To), nv.ptr(wT16), k * C,
                                             _rows(out, To * Co, Co, Bn, To), None, k * C, Co, nv.EPI_NONE, None, None, 0, st))


@pytest.mark.parametrize("M,K1,N", [(4096, 200, 512), (1000, 1536, 512), (256, 3000, 512), (50, 8, 8), (8448, 512, 1504),
                                    (37, 136, 264), (64, 128, 128), (65, 128, 128), (25344, 1536, 512)])
def test_bf16_storage_gemm_tn_and_bias_grad(M, K1, N):
    """lidbox_gemm_bf16s_tn (wgrad on bf16 shadows, transpose-read operands): the float64 product of the stored bf16
    values to fp32 summation round-off; accumulate; deterministic; bias gradient = fp32 column sums of the shadow"""
    from lidbox_amd import _native as nv
    rng = np.random.default_rng(M + N)
    A, B = rng.standard_normal((M, K1)), rng.standard_normal((M, N))
    # a transpose- / permutation-detecting pattern on top of the noise: every row and column gets its own scale
    A *= (1.0 + 0.01 * np.arange(K1))[None, :]
    B *= (1.0 + 0.013 * np.arange(N))[None, :]
    ref = _bf16(A).T @ _bf16(B)
    a16, b16 = _dev(A).bfloat16(), _dev(B).bfloat16()
    c = torch.full((K1, N), 3.0, device="cuda")
    st = nv.current_stream()
    wsb = nv.lib.lidbox_gemm_bf16s_tn_workspace(M, K1, N)
    ws = _ws(wsb)
    ra, rb = nv.Rows(a16.data_ptr(), 0, K1, 1, M), nv.Rows(b16.data_ptr(), 0, N, 1, M)
    nv.check(nv.lib.lidbox_gemm_bf16s_tn(ra, rb, nv.ptr(c), N, K1, N, 0, None, nv.ptr(ws), wsb, st))
    _close(c.cpu().numpy(), ref)
    nv.check(nv.lib.lidbox_gemm_bf16s_tn(ra, rb, nv.ptr(c), N, K1, N, 1, None, nv.ptr(ws), wsb, st))
    _close(c.cpu().numpy(), 2 * ref)
    c2, c3 = torch.empty_like(c), torch.empty_like(c)
    nv.check(nv.lib.lidbox_gemm_bf16s_tn(ra, rb, nv.ptr(c2), N, K1, N, 0, None, nv.ptr(ws), wsb, st))
    bg = torch.full((N,), 9.0, device="cuda")
    nv.check(nv.lib.lidbox_gemm_bf16s_tn(ra, rb, nv.ptr(c3), N, K1, N, 0, nv.ptr(bg), nv.ptr(ws), wsb, st))
    assert torch.equal(c2, c3)
    _close(bg.cpu().numpy(), _bf16(B).sum(axis=0), 1e-5)
    # the fp32-source wgrad kernel on the same (already bf16-representable) values agrees to summation order
    a32, b32 = a16.float(), b16.float()
    wsb2 = nv.lib.lidbox_gemm_bf16_tn_workspace(M, K1, N)
    ws2 = _ws(wsb2)
    c_old = torch.empty_like(c)
    nv.check(nv.lib.lidbox_gemm_bf16_tn(_rows(a32, 0, K1, 1, M), _rows(b32, 0, N, 1, M), nv.ptr(c_old), N, K1, N, 0, None,
                                        nv.ptr(ws2), wsb2, st))
    _close(c2.cpu().numpy(), c_old.cpu().double().numpy(), rel=1e-5)


@pytest.mark.parametrize("Bn,T,C,k,s_,Co", [(5, 21, 16, 3, 2, 24), (3, 7, 8, 5, 1, 8), (40, 3, 32, 1, 1, 136), (2, 400, 64, 3, 3, 128)])
def test_bf16_storage_gemm_tn_implicit_rows_and_errors(Bn, T, C, k, s_, Co):
    """wgrad of a strided causal Conv1D over bf16 shadows: overlapping input windows [B, pad + T, C] (implicit rows) against
    the output-gradient shadow [B, To, Co]; utterances shorter than the 16-row load stride included"""
    from lidbox_amd import _native as nv
    rng = np.random.default_rng(Bn * 100 + T)
    pad = k - 1
    x = np.zeros((Bn, pad + T, C))
    x[:, pad:] = rng.standard_normal((Bn, T, C))
    To = (T - 1) // s_ + 1
    dy = rng.standard_normal((Bn, To, Co))
    idx = np.arange(To)[:, None] * s_ + np.arange(k)[None, :]
    col = _bf16(x)[:, idx, :].reshape(Bn * To, k * C)
    ref = col.T @ _bf16(dy).reshape(Bn * To, Co)
    st = nv.current_stream()
    x16, dy16 = _dev(x).bfloat16(), _dev(dy).bfloat16()
    M, K1 = Bn * To, k * C
    wsb = nv.lib.lidbox_gemm_bf16s_tn_workspace(M, K1, Co)
    ws = _ws(wsb)
    dW = torch.full((K1, Co), -1.0, device="cuda")
    db = torch.zeros(Co, device="cuda")
    ra = nv.Rows(x16.data_ptr(), (pad + T) * C, s_ * C, Bn, To)
    rb = nv.Rows(dy16.data_ptr(), To * Co, Co, Bn, To)
    nv.check(nv.lib.lidbox_gemm_bf16s_tn(ra, rb, nv.ptr(dW), Co, K1, Co, 0, nv.ptr(db), nv.ptr(ws), wsb, st))
    _close(dW.cpu().numpy(), ref)
    _close(db.cpu().numpy(), _bf16(dy).reshape(M, Co).sum(axis=0), 1e-5)
    # N not a multiple of 8 is fine when the rows are padded to one (the last 16-byte piece stays inside the row) ...
    dW4 = torch.full((K1, Co - 4), -1.0, device="cuda")
    nv.check(nv.lib.lidbox_gemm_bf16s_tn(ra, rb, nv.ptr(dW4), Co - 4, K1, Co - 4, 0, nv.ptr(db), nv.ptr(ws), wsb, st))
    _close(dW4.cpu().numpy(), ref[:, :Co - 4])
    _close(db.cpu().numpy()[:Co - 4], _bf16(dy).reshape(M, Co).sum(axis=0)[:Co - 4], 1e-5)
    with pytest.raises(ValueError):                          # ... and refused when they are not
        nv.check(nv.lib.lidbox_gemm_bf16s_tn(ra, nv.Rows(dy16.data_ptr(), 0, Co - 8, 1, M), nv.ptr(dW), Co, K1, Co - 4, 0, None,
                                             nv.ptr(ws), wsb, st))
    with pytest.raises(ValueError):                          # row stride not a multiple of 8 elements
        nv.check(nv.lib.lidbox_gemm_bf16s_tn(nv.Rows(x16.data_ptr(), (pad + T) * C, 12, Bn, To), rb, nv.ptr(dW), Co, K1, Co, 0,
                                             None, nv.ptr(ws), wsb, st))
    with pytest.raises(ValueError):                          # workspace too small
        nv.check(nv.lib.lidbox_gemm_bf16s_tn(ra, rb, nv.ptr(dW), Co, K1, Co, 0, None, nv.ptr(ws), 8, st))


@pytest.mark.parametrize("Bn,T,C,k,s_,Co", [(40, 99, 128, 3, 3, 256),      # utterances of 33 rows: a wrap inside most 32-row steps
                                            (300, 7, 64, 2, 1, 256),       # 7-row utterances: several wraps per step and per piece
                                            (6, 700, 256, 2, 2, 512),      # long utterances, 2 x 2 tiles
                                            (9, 130, 40, 5, 1, 264),       # K1 = 200, N = 264: tiles that hang over both edges
                                            (1, 5000, 512, 1, 1, 512)])    # flat rows, a slice that ends inside a step
def test_bf16_storage_wgrad_pingpong_tile(Bn, T, C, k, s_, Co, monkeypatch):
    """gemm16s_tn_pp_kernel (256 x 256 eight-wave tile of the storage wgrad, LDS-DMA rows + transpose reads), forced on shapes the
    policy would leave to the four-wave kernel: the float64 product of the stored values, the bias gradient off the matrix
    pipe, accumulate, run-to-run identical, and equal to round-off with the four-wave kernel's result"""
    from lidbox_amd import _native as nv
    rng = np.random.default_rng(Bn * 1000 + T)
    pad = k - 1
    x = np.zeros((Bn, pad + T, C))
    x[:, pad:] = rng.standard_normal((Bn, T, C))
    To = (T - 1) // s_ + 1
    dy = rng.standard_normal((Bn, To, Co)) * (1.0 + 0.01 * np.arange(Co))[None, None, :]
    idx = np.arange(To)[:, None] * s_ + np.arange(k)[None, :]
    col = _bf16(x)[:, idx, :].reshape(Bn * To, k * C)
    ref = col.T @ _bf16(dy).reshape(Bn * To, Co)
    st = nv.current_stream()
    x16, dy16 = _dev(x).bfloat16(), _dev(dy).bfloat16()
    M, K1 = Bn * To, k * C
    ra = nv.Rows(x16.data_ptr(), (pad + T) * C, s_ * C, Bn, To)
    rb = nv.Rows(dy16.data_ptr(), To * Co, Co, Bn, To)
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("LIDBOX_GEMM16_TN_PP", mode)
        wsb = nv.lib.lidbox_gemm_bf16s_tn_workspace(M, K1, Co)
        ws = _ws(wsb)
        dW = torch.full((K1, Co), -1.0, device="cuda")
        db = torch.zeros(Co, device="cuda")
        nv.check(nv.lib.lidbox_gemm_bf16s_tn(ra, rb, nv.ptr(dW), Co, K1, Co, 0, nv.ptr(db), nv.ptr(ws), wsb, st))
        assert (nv.lib.lidbox_gemm_bf16s_tn_last_pp() > 0) == (mode == "1")
        _close(dW.cpu().numpy(), ref)
        _close(db.cpu().numpy(), _bf16(dy).reshape(M, Co).sum(axis=0), 1e-5)
        dW2 = dW.clone()
        nv.check(nv.lib.lidbox_gemm_bf16s_tn(ra, rb, nv.ptr(dW2), Co, K1, Co, 1, None, nv.ptr(ws), wsb, st))
        _close(dW2.cpu().numpy(), 2 * ref)
        dW3 = torch.empty_like(dW)
        nv.check(nv.lib.lidbox_gemm_bf16s_tn(ra, rb, nv.ptr(dW3), Co, K1, Co, 0, None, nv.ptr(ws), wsb, st))
        assert torch.equal(dW, dW3)
        out[mode] = dW.cpu().double().numpy()
    _close(out["1"], out["0"], rel=1e-5)


@pytest.mark.parametrize("Bn,T,C,k,s_,Co", [(9, 130, 40, 5, 1, 256),       # frame1's window shape: K1 = 200 (6 full row blocks + 8 rows)
                                            (70, 33, 40, 5, 1, 512),       # more utterances than slices (two per slice), N = 512
                                            (5, 60, 16, 5, 2, 128),        # stride 2: K1 = 80, row stride 32 elements, one column tile
                                            (3, 198, 24, 9, 1, 128),       # K1 = 216: seven blocks, the last one of 24 rows
                                            (1, 5000, 40, 5, 1, 128),      # one long utterance (flat rows)
                                            (130, 7, 8, 3, 1, 128)])       # utterances much shorter than a 64-row stage: several wraps per piece
def test_bf16_storage_wgrad_k1_resident_kernel(Bn, T, C, k, s_, Co, monkeypatch):
    """gemm16s_tn_kres_kernel (round 6: the whole K1 <= 224 extent in one workgroup's accumulators, windows read out of ONE LDS copy of
    the frames through transpose reads with the input's row stride), forced on: the float64 product of the stored values, the bias
    gradient, accumulate, run-to-run identical bits, and equal to round-off with the four-wave kernel's result"""
    from lidbox_amd import _native as nv
    rng = np.random.default_rng(Bn * 1000 + T)
    pad = k - 1
    x = np.zeros((Bn, pad + T, C))
    x[:, pad:] = rng.standard_normal((Bn, T, C))
    To = (T - 1) // s_ + 1
    dy = rng.standard_normal((Bn, To, Co)) * (1.0 + 0.01 * np.arange(Co))[None, None, :]
    idx = np.arange(To)[:, None] * s_ + np.arange(k)[None, :]
    col = _bf16(x)[:, idx, :].reshape(Bn * To, k * C)
    ref = col.T @ _bf16(dy).reshape(Bn * To, Co)
    st = nv.current_stream()
    x16, dy16 = _dev(x).bfloat16(), _dev(dy).bfloat16()
    M, K1 = Bn * To, k * C
    ra = nv.Rows(x16.data_ptr(), (pad + T) * C, s_ * C, Bn, To)
    rb = nv.Rows(dy16.data_ptr(), To * Co, Co, Bn, To)
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("LIDBOX_GEMM16_TN_KRES", mode)
        monkeypatch.setenv("LIDBOX_GEMM16_TN_PP", "0")
        wsb = nv.lib.lidbox_gemm_bf16s_tn_workspace(M, K1, Co)
        ws = _ws(wsb)
        dW = torch.full((K1, Co), -1.0, device="cuda")
        db = torch.zeros(Co, device="cuda")
        nv.check(nv.lib.lidbox_gemm_bf16s_tn(ra, rb, nv.ptr(dW), Co, K1, Co, 0, nv.ptr(db), nv.ptr(ws), wsb, st))
        assert (nv.lib.lidbox_gemm_bf16s_tn_last_kres() > 0) == (mode == "1")
        _close(dW.cpu().numpy(), ref)
        _close(db.cpu().numpy(), _bf16(dy).reshape(M, Co).sum(axis=0), 1e-5)
        dW2 = dW.clone()
        nv.check(nv.lib.lidbox_gemm_bf16s_tn(ra, rb, nv.ptr(dW2), Co, K1, Co, 1, None, nv.ptr(ws), wsb, st))
        _close(dW2.cpu().numpy(), 2 * ref)
        dW3 = torch.empty_like(dW)
        nv.check(nv.lib.lidbox_gemm_bf16s_tn(ra, rb, nv.ptr(dW3), Co, K1, Co, 0, None, nv.ptr(ws), wsb, st))
        assert torch.equal(dW, dW3)
        out[mode] = dW.cpu().double().numpy()
    _close(out["1"], out["0"], rel=1e-5)


def test_bf16_storage_wgrad_k1_resident_policy_and_fallback(monkeypatch):
    """the K1-resident wgrad runs by default on frame1's launch of configs[4]'s bf16 step (K1 = 200, N = 512, >= 64 k rows) and nowhere else;
    operands it cannot flatten (batch stride not a whole number of row strides, N not in 128-column tiles, K1 > 224) fall back to the
    four-wave kernel even when it is forced, with the same result"""
    from lidbox_amd import _native as nv
    monkeypatch.delenv("LIDBOX_GEMM16_TN_KRES", raising=False)
    st = nv.current_stream()
    rng = np.random.default_rng(5)

    def run(Bn, T, C, k, s_, Co, extra_rows=0, want=None):
        pad = k - 1
        Tp = pad + T + extra_rows
        x = np.zeros((Bn, Tp, C))
        x[:, pad:pad + T] = rng.standard_normal((Bn, T, C))
        To = (T - 1) // s_ + 1
        dy = rng.standard_normal((Bn, To, Co))
        idx = np.arange(To)[:, None] * s_ + np.arange(k)[None, :]
        ref = _bf16(x)[:, idx, :].reshape(Bn * To, k * C).T @ _bf16(dy).reshape(Bn * To, Co)
        x16, dy16 = _dev(x).bfloat16(), _dev(dy).bfloat16()
        M, K1 = Bn * To, k * C
        wsb = nv.lib.lidbox_gemm_bf16s_tn_workspace(M, K1, Co)
        ws = _ws(wsb)
        dW = torch.empty((K1, Co), device="cuda")
        nv.check(nv.lib.lidbox_gemm_bf16s_tn(nv.Rows(x16.data_ptr(), Tp * C, s_ * C, Bn, To), nv.Rows(dy16.data_ptr(), To * Co, Co, Bn, To),
                                             nv.ptr(dW), Co, K1, Co, 0, None, nv.ptr(ws), wsb, st))
        _close(dW.cpu().numpy(), ref)
        if want is not None:
            assert (nv.lib.lidbox_gemm_bf16s_tn_last_kres() > 0) == want, (Bn, T, C, k, s_, Co)

    run(352, 198, 40, 5, 1, 512, want=True)           # frame1 at 352 utterances: 69 696 rows
    run(256, 198, 40, 5, 1, 512, want=False)          # 50 688 rows: below the policy's floor (measured slower inside the step)
    run(352, 198, 48, 3, 1, 512, want=True)           # K1 = 144, row stride 48: one stage of frames is 3 168 elements (<= 4 096)
    run(352, 198, 64, 3, 1, 512, want=False)          # K1 = 192, row stride 64: 4 224 elements do not fit one piece per thread
    run(64, 99, 512, 3, 2, 512, want=False)           # frame2: K1 = 1536
    monkeypatch.setenv("LIDBOX_GEMM16_TN_KRES", "1")
    run(6, 50, 40, 5, 2, 128, extra_rows=1, want=False)      # 55 padded rows of 40, row stride 80: batch stride not a multiple
    run(6, 50, 40, 5, 1, 136, want=False)                    # N = 136
    run(6, 50, 48, 5, 1, 128, want=False)                    # K1 = 240
    run(6, 50, 40, 5, 1, 128, want=True)


def test_bf16_storage_wgrad_pingpong_policy_and_fallback(monkeypatch):
    """the ping-pong wgrad tile runs where it was measured faster (frame2's wgrad at 512 utterances: 12 tiles x 21 slices of 2 432
    rows) and nowhere else by default; operands it cannot walk with one utterance counter fall back even when it is forced"""
    from lidbox_amd import _native as nv
    monkeypatch.delenv("LIDBOX_GEMM16_TN_PP", raising=False)
    st = nv.current_stream()

    def run(M, K1, N, batched_b=None):
        a16 = torch.randn(M, K1, device="cuda").bfloat16()
        b16 = torch.randn(M, N, device="cuda").bfloat16()
        wsb = nv.lib.lidbox_gemm_bf16s_tn_workspace(M, K1, N)
        ws = _ws(wsb)
        c = torch.empty(K1, N, device="cuda")
        ra = nv.Rows(a16.data_ptr(), 0, K1, 1, M)
        rb = nv.Rows(b16.data_ptr(), 0, N, 1, M) if batched_b is None else nv.Rows(b16.data_ptr(), batched_b * N, N, M // batched_b, batched_b)
        nv.check(nv.lib.lidbox_gemm_bf16s_tn(ra, rb, nv.ptr(c), N, K1, N, 0, None, nv.ptr(ws), wsb, st))
        torch.cuda.synchronize()
        ref = a16.double().T @ b16.double()
        assert float((c.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
        return nv.lib.lidbox_gemm_bf16s_tn_last_pp()

    assert run(512 * 99, 1536, 512) == 21
    assert run(256 * 99, 1536, 512) == 0          # 1 216-row slices: the partial sums outweigh the faster loop
    assert run(512 * 33, 512, 512) == 0           # 4 tiles
    monkeypatch.setenv("LIDBOX_GEMM16_TN_PP", "1")
    assert run(8448, 512, 512) > 0
    assert run(8448, 512, 512, batched_b=33) == 0  # A flat, B in utterances of 33 rows: the four-wave kernel


@pytest.mark.parametrize("Bn,T,C,k,s_,N", [(6, 198, 40, 5, 1, 512),      # frame1 of the x-vector
                                           (3, 33, 16, 3, 1, 136),        # a column tile that hangs over N, 33-row utterances
                                           (5, 256, 8, 5, 1, 128),        # utterances as long as the tile
                                           (4, 100, 8, 5, 2, 264),        # strided windows: K = 40, 50 rows per utterance
                                           (2, 70, 24, 3, 1, 128)])       # K = 72: a half k slice at the end
def test_bf16_storage_short_contraction_resident_kernel(Bn, T, C, k, s_, N, monkeypatch):
    """gemm16s_rows_kres_kernel (gemm16_kres.h: a 64-column weight panel and, per wave, the frames of 32 rows resident in LDS,
    implicit rows read out of the frame image), forced on: float64 product of the stored values + bias + ReLU, fp32 output and shadow, shadow only, epilogue
    variants; equal to round-off with the LDS-DMA tiles; refused shapes fall back"""
    import ctypes
    from lidbox_amd import _native as nv
    rng = np.random.default_rng(Bn * 100 + T)
    pad = k - 1
    x = np.zeros((Bn, pad + T, C))
    x[:, pad:] = rng.standard_normal((Bn, T, C))
    To = (T - 1) // s_ + 1
    K = k * C
    w = rng.standard_normal((N, K)) * 0.1 * (1.0 + 0.01 * np.arange(K))[None, :]
    bias = rng.standard_normal(N)
    idx = np.arange(To)[:, None] * s_ + np.arange(k)[None, :]
    col = _bf16(x)[:, idx, :].reshape(Bn * To, K)
    pre = col @ _bf16(w).T
    st = nv.current_stream()
    x16, w16, bd = _dev(x).bfloat16(), _dev(w).bfloat16(), _dev(bias)
    ra = nv.Rows(x16.data_ptr(), (pad + T) * C, s_ * C, Bn, To)
    wsb = max(16, nv.lib.lidbox_gemm_bf16_rows_workspace(Bn * To, N, K))
    ws = _ws(wsb)
    v = (ctypes.c_int * 3)()
    outs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("LIDBOX_GEMM16S_KRES", mode)
        for epi, ref in ((nv.EPI_BIAS_RELU, np.maximum(pre + bias, 0)), (nv.EPI_BIAS, pre + bias), (nv.EPI_NONE, pre), (nv.EPI_RELU, np.maximum(pre, 0))):
            c32 = torch.full((Bn, To, N), 7.0, device="cuda")
            c16 = torch.zeros((Bn, To, N), dtype=torch.bfloat16, device="cuda")
            nv.check(nv.lib.lidbox_gemm_bf16s_nt(ra, nv.ptr(w16), K, _rows(c32, To * N, N, Bn, To), nv.ptr(c16), K, N, epi,
                                                 nv.ptr(bd) if epi in (nv.EPI_BIAS_RELU, nv.EPI_BIAS) else None, nv.ptr(ws), wsb, st))
            nv.check(nv.lib.lidbox_gemm_bf16s_last_variant(v))
            assert (list(v) == [1, 64, 1]) == (mode == "1"), list(v)
            _close(c32.cpu().numpy().reshape(-1, N), ref)
            assert torch.equal(c16, c32.bfloat16())
            outs[(mode, epi)] = c32.cpu().double().numpy()
        # shadow only, into rows wider than N (a padded shadow) through a flat descriptor
        Np = N + 8
        sh = torch.full((Bn * To, Np), 3.0, dtype=torch.bfloat16, device="cuda")
        nv.check(nv.lib.lidbox_gemm_bf16s_nt(ra, nv.ptr(w16), K, nv.Rows(None, 0, Np, 1, Bn * To), nv.ptr(sh), K, N, nv.EPI_BIAS_RELU, nv.ptr(bd),
                                             nv.ptr(ws), wsb, st))
        nv.check(nv.lib.lidbox_gemm_bf16s_last_variant(v))
        assert (list(v) == [1, 64, 1]) == (mode == "1")
        assert torch.equal(sh[:, :N], torch.from_numpy(outs[(mode, nv.EPI_BIAS_RELU)]).float().cuda().reshape(-1, N).bfloat16())
        assert bool((sh[:, N:] == 3.0).all())
    for epi in (nv.EPI_BIAS_RELU, nv.EPI_NONE):
        _close(outs[("1", epi)], outs[("0", epi)], rel=1e-5)
    # flat rows (one "utterance" of 300 rows, windows that do not overlap) run on it when their 32-row image fits; a ReLU-mask
    # epilogue (dgrad) is not its business
    monkeypatch.setenv("LIDBOX_GEMM16S_KRES", "1")
    a16 = _dev(rng.standard_normal((300, K))).bfloat16()
    c32 = torch.empty((300, N), device="cuda")
    ra2, rc2 = nv.Rows(a16.data_ptr(), 0, K, 1, 300), _rows(c32, 0, N, 1, 300)
    nv.check(nv.lib.lidbox_gemm_bf16s_nt(ra2, nv.ptr(w16), K, rc2, None, K, N, nv.EPI_NONE, None, nv.ptr(ws), wsb, st))
    nv.check(nv.lib.lidbox_gemm_bf16s_last_variant(v))
    assert (list(v) == [1, 64, 1]) == (31 * K * 2 + (K + 15) // 16 * 32 <= 3072)
    ref2 = _bf16(a16.float().cpu().numpy()) @ _bf16(w).T
    _close(c32.cpu().numpy(), ref2)
    mk = _dev(rng.standard_normal((300, N)))
    nv.check(nv.lib.lidbox_gemm_bf16s_nt(ra2, nv.ptr(w16), K, rc2, None, K, N, nv.EPI_RELU_MASK, nv.ptr(mk), nv.ptr(ws), wsb, st))
    nv.check(nv.lib.lidbox_gemm_bf16s_last_variant(v))
    assert list(v) != [1, 64, 1]
    _close(c32.cpu().numpy(), ref2 * (mk.cpu().numpy() > 0))


def test_bf16_storage_short_contraction_policy(monkeypatch):
    """frame1's forward (K = 200 over windows five frames deep) takes the K-resident kernel on its own when there are utterances
    enough to keep every CU's twelve waves busy (>= 2 x 256 column-tile x utterance pairs); small batches stay on the LDS-DMA tiles"""
    import ctypes
    from lidbox_amd import _native as nv
    monkeypatch.delenv("LIDBOX_GEMM16S_KRES", raising=False)
    st = nv.current_stream()
    v = (ctypes.c_int * 3)()
    for Bn, want in ((256, True), (64, True), (16, False)):
        x16 = torch.randn(Bn, 202, 40, device="cuda").bfloat16()
        w16 = (torch.randn(512, 200, device="cuda") * 0.05).bfloat16()
        c16 = torch.empty(Bn, 198, 512, dtype=torch.bfloat16, device="cuda")
        wsb = max(16, nv.lib.lidbox_gemm_bf16_rows_workspace(Bn * 198, 512, 200))
        ws = _ws(wsb)
        nv.check(nv.lib.lidbox_gemm_bf16s_nt(nv.Rows(x16.data_ptr(), 202 * 40, 40, Bn, 198), nv.ptr(w16), 200, nv.Rows(None, 198 * 512, 512, Bn, 198),
                                             nv.ptr(c16), 200, 512, nv.EPI_RELU, None, nv.ptr(ws), wsb, st))
        nv.check(nv.lib.lidbox_gemm_bf16s_last_variant(v))
        assert (list(v) == [1, 64, 1]) == want, (Bn, list(v))
        col = x16.float().unfold(1, 5, 1).permute(0, 1, 3, 2).reshape(Bn * 198, 200)         # [b, t, tap, channel]
        ref = torch.relu(col.double() @ w16.double().T)
        assert float((c16.double().reshape(-1, 512) - ref).abs().max()) <= 4e-3 * float(ref.abs().max())


def test_refresh_bf16_weights_one_launch():
    """flat -> flat16 plus bf16 images of listed matrices inside it (the per-step weight-shadow refresh): transposed,
    row-padded, and blocks side by side in one destination"""
    from lidbox_amd import _native as nv
    rng = np.random.default_rng(5)
    shapes = [(200, 512), (1536, 512), (37, 5), (512, 1500), (132, 68)]       # the last: vector path with partial 64 x 64 tiles
    offs, off = [], 3 * 4                                       # matrices start on 4-float boundaries inside the flat vector
    for r, c in shapes:
        offs.append(off)
        off += (r * c + 3) // 4 * 4 + 8
    n = off + 5
    flat = _dev(rng.standard_normal(n))
    flat16 = torch.zeros(n, dtype=torch.bfloat16, device="cuda")
    dsts = [torch.zeros((c, r), dtype=torch.bfloat16, device="cuda") for r, c in shapes]
    padded = torch.zeros((512, 1504), dtype=torch.bfloat16, device="cuda")             # (512, 1500) with rows padded to 1504
    stacked = torch.zeros((200, 2 * 512), dtype=torch.bfloat16, device="cuda")          # (200, 512) twice, side by side
    items = [(o, r, c, d.data_ptr(), r, 1) for (r, c), o, d in zip(shapes, offs, dsts)]
    items.append((offs[3], 512, 1500, padded.data_ptr(), 1504, 0))
    items.append((offs[0], 200, 512, stacked.data_ptr() + 2 * 512, 1024, 0))
    items.append((offs[0], 200, 512, stacked.data_ptr(), 1024, 0))
    odd = torch.zeros((6, 7), dtype=torch.bfloat16, device="cuda")                      # source off the 4-float grid, odd row pitch
    items.append((offs[2] + 1, 6, 5, odd.data_ptr(), 7, 0))
    odd_t = torch.zeros((68, 134), dtype=torch.bfloat16, device="cuda")                 # transposed at a row pitch that is no multiple of 4
    items.append((offs[4], 132, 68, odd_t.data_ptr(), 134, 1))
    mats = (nv.WeightShadow * len(items))(*[nv.WeightShadow(*it) for it in items])
    st = nv.current_stream()
    nv.check(nv.lib.lidbox_refresh_bf16_weights(nv.ptr(flat), nv.ptr(flat16), n, mats, len(items), st))
    assert torch.equal(flat16, flat.bfloat16())
    assert torch.equal(odd[:, :5], flat[offs[2] + 1:offs[2] + 31].reshape(6, 5).bfloat16()) and not odd[:, 5:].any()
    assert torch.equal(odd_t[:, :132], flat[offs[4]:offs[4] + 132 * 68].reshape(132, 68).t().bfloat16()) and not odd_t[:, 132:].any()
    for (r, c), o, d in zip(shapes, offs, dsts):
        assert torch.equal(d, flat[o:o + r * c].reshape(r, c).t().bfloat16())
    w3 = flat[offs[3]:offs[3] + 512 * 1500].reshape(512, 1500).bfloat16()
    assert torch.equal(padded[:, :1500], w3) and not padded[:, 1500:].any()
    w0 = flat[offs[0]:offs[0] + 200 * 512].reshape(200, 512).bfloat16()
    assert torch.equal(stacked[:, :512], w0) and torch.equal(stacked[:, 512:], w0)
    # more matrices than one launch's table holds
    many = (nv.WeightShadow * 100)(*[nv.WeightShadow(*items[i % 5]) for i in range(100)])
    for d in dsts:
        d.zero_()
    nv.check(nv.lib.lidbox_refresh_bf16_weights(nv.ptr(flat), nv.ptr(flat16), n, many, 100, st))
    for (r, c), o, d in zip(shapes, offs, dsts):
        assert torch.equal(d, flat[o:o + r * c].reshape(r, c).t().bfloat16())
    nv.check(nv.lib.lidbox_refresh_bf16_weights(nv.ptr(flat), nv.ptr(flat16), n, None, 0, st))      # no matrices: convert only
    # flat16 == NULL: only the listed matrices -- one of them an in-place image inside a flat16 buffer that is otherwise left alone
    part16 = torch.full((n,), 3.0, dtype=torch.bfloat16, device="cuda")
    for d in dsts:
        d.zero_()
    some = items[:5] + [(offs[1], shapes[1][0], shapes[1][1], part16.data_ptr() + 2 * offs[1], shapes[1][1], 0)]
    some_c = (nv.WeightShadow * len(some))(*[nv.WeightShadow(*it) for it in some])
    nv.check(nv.lib.lidbox_refresh_bf16_weights(nv.ptr(flat), None, n, some_c, len(some), st))
    for (r, c), o, d in zip(shapes, offs, dsts):
        assert torch.equal(d, flat[o:o + r * c].reshape(r, c).t().bfloat16())
    r1, c1 = shapes[1]
    assert torch.equal(part16[offs[1]:offs[1] + r1 * c1], flat[offs[1]:offs[1] + r1 * c1].bfloat16())
    assert (part16[:offs[1]] == 3.0).all() and (part16[offs[1] + r1 * c1:] == 3.0).all()
    with pytest.raises(ValueError):                              # a matrix that sticks out of the vector
        mats[0] = nv.WeightShadow(n - 10, 200, 512, dsts[0].data_ptr(), 200, 1)
        nv.check(nv.lib.lidbox_refresh_bf16_weights(nv.ptr(flat), nv.ptr(flat16), n, mats, len(items), st))
    with pytest.raises(ValueError):                              # destination rows shorter than the data
        mats[0] = nv.WeightShadow(offs[0], 200, 512, dsts[0].data_ptr(), 100, 1)
        nv.check(nv.lib.lidbox_refresh_bf16_weights(nv.ptr(flat), nv.ptr(flat16), n, mats, len(items), st))


@pytest.mark.parametrize("M,K,N", [(700, 512, 1024), (33, 1536, 512), (8448 // 4, 1504, 512)])
def test_bf16_storage_gemm_shadow_only_output_and_bf16_mask(M, K, N):
    """C.base == NULL: the result exists only as the shadow; LIDBOX_EPI_MASK_BF16: the ReLU mask is read from bf16 data.
    Same numbers as the launch that also writes fp32 and reads a fp32 mask -- including the split / tail-split plans"""
    from lidbox_amd import _native as nv
    rng = np.random.default_rng(M + K)
    a16, b16 = _dev(rng.standard_normal((M, K))).bfloat16(), _dev(rng.standard_normal((N, K))).bfloat16()
    mask32 = _dev(np.maximum(rng.standard_normal((M, N)), 0))          # a ReLU output: zeros and positives
    mask16 = mask32.bfloat16()
    st = nv.current_stream()
    ra = nv.Rows(a16.data_ptr(), 0, K, 1, M)
    wsb = max(16, nv.lib.lidbox_gemm_bf16_rows_workspace(M, N, K))
    ws = _ws(wsb)
    for epi, aux in ((nv.EPI_NONE, None), (nv.EPI_RELU_MASK, mask32)):
        c = torch.zeros((M, N), device="cuda")
        c16 = torch.zeros((M, N), dtype=torch.bfloat16, device="cuda")
        nv.check(nv.lib.lidbox_gemm_bf16s_nt(ra, nv.ptr(b16), K, _rows(c, 0, N, 1, M), nv.ptr(c16), K, N, epi, nv.ptr(aux),
                                             nv.ptr(ws), wsb, st))
        only = torch.full((M, N), 5.0, dtype=torch.bfloat16, device="cuda")
        epi2, aux2 = (epi | nv.EPI_MASK_BF16, mask16) if aux is not None else (epi, None)
        nv.check(nv.lib.lidbox_gemm_bf16s_nt(ra, nv.ptr(b16), K, nv.Rows(None, 0, N, 1, M), nv.ptr(only), K, N, epi2, nv.ptr(aux2),
                                             nv.ptr(ws), wsb, st))
        assert torch.equal(only, c16) and torch.equal(c16, c.bfloat16())
    with pytest.raises(ValueError):                              # no fp32 C and no shadow
        nv.check(nv.lib.lidbox_gemm_bf16s_nt(ra, nv.ptr(b16), K, nv.Rows(None, 0, N, 1, M), None, K, N, nv.EPI_NONE, None, None, 0, st))
    with pytest.raises(ValueError):                              # accumulating epilogues read C
        nv.check(nv.lib.lidbox_gemm_bf16s_nt(ra, nv.ptr(b16), K, nv.Rows(None, 0, N, 1, M), nv.ptr(only), K, N, nv.EPI_ACCUM, None,
                                             None, 0, st))
    with pytest.raises(ValueError):                              # the flag without a mask epilogue
        nv.check(nv.lib.lidbox_gemm_bf16s_nt(ra, nv.ptr(b16), K, nv.Rows(None, 0, N, 1, M), nv.ptr(only), K, N,
                                             nv.EPI_BIAS | nv.EPI_MASK_BF16, nv.ptr(mask16), None, 0, st))


@pytest.mark.parametrize("M,K,N,want", [(8448, 512, 512, [64, 128, 2]), (8448, 512, 1536, [256, 256, 2]), (25344, 1024, 512, [256, 256, 2]),
                                         (25344, 1536, 512, [256, 256, 2]), (50688, 200, 512, [64, 128, 2]), (16896, 1504, 512, [256, 256, 2])])
def test_bf16_storage_gemm_policy_at_layer_sizes(M, K, N, want):
    """x-vector layer shapes at bs 256: which kernel lidbox_gemm_bf16s_nt picks on its own (csrc/gemm_bf16.hip:
    choose_dma16 -- the eight-wave 256 x 256 ping-pong tile where its tiles fill three quarters of the CUs and K >= 512, or half
    of them and K >= 1500; 64 x 128 LDS-DMA tiles for quarter rounds and short contractions) and that the product is the
    bf16-operand product whichever it is"""
    from lidbox_amd import _native as nv
    rng = np.random.default_rng(K + N)
    A, B = rng.standard_normal((M, K)).astype(np.float32), rng.standard_normal((N, K)).astype(np.float32)
    a16, b16 = torch.from_numpy(A).cuda().bfloat16(), torch.from_numpy(B).cuda().bfloat16()
    mask = torch.from_numpy(rng.standard_normal((M, N)).astype(np.float32)).cuda()
    c = torch.full((M, N), 1.0, device="cuda")
    c16 = torch.zeros((M, N), dtype=torch.bfloat16, device="cuda")
    st = nv.current_stream()
    wsb = max(16, nv.lib.lidbox_gemm_bf16_rows_workspace(M, N, K))
    ws = _ws(wsb)
    nv.check(nv.lib.lidbox_gemm_bf16s_nt(nv.Rows(a16.data_ptr(), 0, K, 1, M), nv.ptr(b16), K, _rows(c, 0, N, 1, M), nv.ptr(c16), K, N,
                                         nv.EPI_ACCUM_RELU_MASK, nv.ptr(mask), nv.ptr(ws), wsb, st))
    out3 = (nv.C.c_int * 3)()
    nv.check(nv.lib.lidbox_gemm_bf16s_last_variant(out3))
    assert list(out3) == want
    ref = (a16.float().double() @ b16.float().double().T) * (mask > 0) + 1.0          # float64 product of the bf16 operands, on the GPU
    err = float((c.double() - ref).abs().max()) / float(ref.abs().max())
    assert err < 2e-5, err
    assert torch.equal(c16, c.bfloat16())
