"""Diagnostic: NN GEMM with an artificial A row stride (overlapping implicit rows) -- same instruction
stream and byte count per workgroup, different footprint.  usage: gemm_rowstride.py M K N stride_floats [--bf16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lidbox_amd import _native as nv
M, K, N, rs = (int(a) for a in sys.argv[1:5])
fn = nv.lib.lidbox_gemm_bf16_nn if "--bf16" in sys.argv else nv.lib.lidbox_gemm_nn
st = nv.current_stream()
x = torch.randn(M * rs + K + 64, device="cuda")
W = torch.randn(K, N, device="cuda")
y = torch.zeros(M, N, device="cuda")
f = lambda: nv.check(fn(nv.Rows(x.data_ptr(), 0, rs, 1, M), nv.ptr(W), N, nv.Rows(y.data_ptr(), 0, N, 1, M), K, N, 0, None, None, 0, st))
f(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): f()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 200
print("M=%d K=%d N=%d row_stride=%d floats: %.1f us %.1f TF/s" % (M, K, N, rs, us, 2.0 * M * K * N / us / 1e6))
