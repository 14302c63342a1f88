import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from lidbox_amd import _native as nv
def dev(x): return torch.from_numpy(np.ascontiguousarray(np.asarray(x, np.float32))).cuda()
def rows(t, bs, rs, batch, rpb): return nv.Rows(t.data_ptr(), bs, rs, batch, rpb)
rng = np.random.default_rng(0)
st = nv.current_stream()
ok = True
for (M, K, N, epi) in [(8448, 512, 512, "mask"), (8448, 1500, 512, "none"), (25344, 512, 1024, "mask"), (4100, 516, 200, "accmask"), (8448, 512, 1536, "mask")]:
    A, W = rng.standard_normal((M, K)), rng.standard_normal((N, K)) * 0.1
    mask, old = rng.standard_normal((M, N)), rng.standard_normal((M, N))
    a, w, mk = dev(A), dev(W), dev(mask)
    ws = torch.empty(max(16, nv.lib.lidbox_gemm_rows_workspace(M, N, K)) + 1024, dtype=torch.uint8, device="cuda"); ws.fill_(0xAB)
    e = {"mask": nv.EPI_RELU_MASK, "none": nv.EPI_NONE, "accmask": nv.EPI_ACCUM_RELU_MASK}[epi]
    ref = A @ W.T
    if epi != "none": ref = ref * (mask > 0)
    if epi == "accmask": ref = ref + old.astype(np.float32).astype(np.float64)
    outs = {}
    for mode in ("", "64", "128"):
        if mode: os.environ["LIDBOX_GEMM_NT8"] = mode
        else: os.environ.pop("LIDBOX_GEMM_NT8", None)
        c = dev(old).clone() if epi == "accmask" else torch.full((M, N), 9.0, device="cuda")
        nv.check(nv.lib.lidbox_gemm_nt(rows(a, 0, K, 1, M), nv.ptr(w), K, rows(c, 0, N, 1, M), K, N, e, nv.ptr(mk) if epi != "none" else None, nv.ptr(ws), ws.numel(), st))
        torch.cuda.synchronize()
        fam = nv.lib.lidbox_gemm_last_family()
        err = float(np.abs(c.cpu().numpy() - ref).max() / np.abs(ref).max())
        outs[mode] = c
        # timing
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3): nv.lib.lidbox_gemm_nt(rows(a, 0, K, 1, M), nv.ptr(w), K, rows(c, 0, N, 1, M), K, N, nv.EPI_RELU_MASK if epi != "none" else nv.EPI_NONE, nv.ptr(mk) if epi != "none" else None, nv.ptr(ws), ws.numel(), st)
        e0.record()
        for _ in range(20): nv.lib.lidbox_gemm_nt(rows(a, 0, K, 1, M), nv.ptr(w), K, rows(c, 0, N, 1, M), K, N, nv.EPI_RELU_MASK if epi != "none" else nv.EPI_NONE, nv.ptr(mk) if epi != "none" else None, nv.ptr(ws), ws.numel(), st)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print("M=%d K=%d N=%d %-8s mode=%-4s family=%d rel err %.2e  %.1f us  %.1f TFLOP/s" % (M, K, N, epi, mode or "dflt", fam, err, us, 2.0 * M * K * N / us / 1e6))
        ok &= err < 2e-5
print("OK" if ok else "FAILED")
