"""correctness + isolated timing of the eight-wave forward tiles (LIDBOX_GEMM_NN8) against the default plan"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidbox_amd import _native as nv
def dev(x): return torch.from_numpy(np.ascontiguousarray(np.asarray(x, np.float32))).cuda()
def rows(t, bs, rs, batch, rpb): return nv.Rows(t.data_ptr(), bs, rs, batch, rpb)
rng = np.random.default_rng(0)
st = nv.current_stream()
ok = True
for (M, K, N) in [(50688, 200, 512), (8448, 1536, 512), (8448, 512, 512), (8448, 512, 1500), (4100, 516, 200)]:
    A, W, b = rng.standard_normal((M, K)), rng.standard_normal((K, N)) * 0.1, rng.standard_normal(N)
    a, w, bias = dev(A), dev(W), dev(b)
    ws = torch.empty(max(16, nv.lib.lidbox_gemm_rows_workspace(M, N, K)) + 1024, dtype=torch.uint8, device="cuda"); ws.fill_(0xAB)
    ref = np.maximum(A @ W + b, 0.0)
    for mode in ("", "64", "128"):
        if mode: os.environ["LIDBOX_GEMM_NN8"] = mode
        else: os.environ.pop("LIDBOX_GEMM_NN8", None)
        c = torch.full((M, N), 9.0, device="cuda")
        call = lambda: nv.lib.lidbox_gemm_nn(rows(a, 0, K, 1, M), nv.ptr(w), N, rows(c, 0, N, 1, M), K, N, nv.EPI_BIAS_RELU, nv.ptr(bias), nv.ptr(ws), ws.numel(), st)
        nv.check(call()); torch.cuda.synchronize()
        fam = nv.lib.lidbox_gemm_last_family()
        err = float(np.abs(c.cpu().numpy() - ref).max() / np.abs(ref).max())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3): call()
        e0.record()
        for _ in range(20): call()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print("nn M=%d K=%d N=%d mode=%-4s family=%d rel err %.2e  %.1f us  %.1f TFLOP/s" % (M, K, N, mode or "dflt", fam, err, us, 2.0 * M * K * N / us / 1e6))
        ok &= err < 2e-5
print("OK" if ok else "FAILED")
