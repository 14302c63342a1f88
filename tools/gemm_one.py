"""One GEMM shape, repeated: for PMC collection.  usage: gemm_one.py nn|nt|tn M K N [--bf16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lidbox_amd import _native as nv
kind, M, K, N = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
_P = "lidbox_gemm_bf16_" if "--bf16" in sys.argv else "lidbox_gemm_"
GEMM_NN, GEMM_NT, GEMM_TN = (getattr(nv.lib, _P + n) for n in ("nn", "nt", "tn"))
st = nv.current_stream()
ws = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
if kind == "nn":
    x = torch.randn(M, K, device="cuda"); W = torch.randn(K, N, device="cuda"); y = torch.zeros(M, N, device="cuda")
    f = lambda: nv.check(GEMM_NN(nv.Rows(x.data_ptr(), 0, K, 1, M), nv.ptr(W), N, nv.Rows(y.data_ptr(), 0, N, 1, M), K, N, 0, None, nv.ptr(ws), ws.numel(), st))
elif kind == "nt":
    x = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda"); y = torch.zeros(M, N, device="cuda")
    f = lambda: nv.check(GEMM_NT(nv.Rows(x.data_ptr(), 0, K, 1, M), nv.ptr(W), K, nv.Rows(y.data_ptr(), 0, N, 1, M), K, N, 0, None, nv.ptr(ws), ws.numel(), st))
else:
    x = torch.randn(M, K, device="cuda"); dy = torch.randn(M, N, device="cuda"); dW = torch.zeros(K, N, device="cuda")
    f = lambda: nv.check(GEMM_TN(nv.Rows(x.data_ptr(), 0, K, 1, M), nv.Rows(dy.data_ptr(), 0, N, 1, M), nv.ptr(dW), N, K, N, 0, None, nv.ptr(ws), ws.numel(), st))
f(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): f()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 200
print("%s M=%d K=%d N=%d: %.1f us %.1f TF/s" % (kind, M, K, N, us, 2.0 * M * K * N / us / 1e6))
