"""frame1's forward as the bf16 step issues it (implicit conv rows over [B, 4 + 198, 40] bf16, shadow-only output, bias + ReLU):
the K-resident kernel (LIDBOX_GEMM16S_KRES=1, gemm16_kres.h) against the policy's LDS-DMA tiles (=0).  usage: python tools/scratch/kres_time.py [B=256]"""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes
import torch
from lidbox_amd import _native as nv
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T, C, k, N = 198, 40, 5, 512
pad, K = k - 1, k * C
st = nv.current_stream()
x16 = torch.zeros(B, pad + T, C, device="cuda").bfloat16(); x16[:, pad:] = torch.randn(B, T, C, device="cuda").bfloat16()
w16 = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
bias = torch.randn(N, device="cuda")
res = {}
for keep32 in (False, True):
    for mode in ("0", "1"):
        os.environ["LIDBOX_GEMM16S_KRES"] = mode
        c16 = torch.empty(B, T, N, dtype=torch.bfloat16, device="cuda")
        c32 = torch.empty(B, T, N, device="cuda") if keep32 else None
        wsb = max(16, nv.lib.lidbox_gemm_bf16_rows_workspace(B * T, N, K)); ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
        ra = nv.Rows(x16.data_ptr(), (pad + T) * C, C, B, T)
        rc = nv.Rows(c32.data_ptr() if keep32 else None, T * N, N, B, T)
        f = lambda: nv.check(nv.lib.lidbox_gemm_bf16s_nt(ra, nv.ptr(w16), K, rc, nv.ptr(c16), K, N, nv.EPI_BIAS_RELU, nv.ptr(bias), nv.ptr(ws), wsb, st))
        f(); torch.cuda.synchronize()
        v = (ctypes.c_int * 3)(); nv.lib.lidbox_gemm_bf16s_last_variant(v)
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): f()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 5 * 1e3)
        res[(keep32, mode)] = c16.float().clone()
        print("B %4d fp32 copy %-5s KRES=%s variant %s  %7.1f us  (%.0f TFLOP/s, output %.0f MB)" % (
            B, keep32, mode, list(v), statistics.median(ts), 2.0 * B * T * K * N / statistics.median(ts) * 1e-6,
            B * T * N * (6 if keep32 else 2) / 1e6), flush=True)
    d = (res[(keep32, "0")] - res[(keep32, "1")]).abs().max() / res[(keep32, "0")].abs().max()
    print("   max |diff| / max = %.2e" % float(d))
