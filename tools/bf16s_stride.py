"""bf16-storage GEMM throughput against the operand row stride (L2 channel spread).  usage: bf16s_stride.py"""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lidbox_amd import _native as nv
st = nv.current_stream()
M, N, K = 25344, 512, 1536
for lda in (1536, 1544, 1568, 1600, 1664, 2048, 1536 + 72):
    for ldb in (K, lda):
        a16 = torch.randn(M, lda, device="cuda").bfloat16(); b16 = torch.randn(N, ldb, device="cuda").bfloat16()
        c = torch.empty(M, N, device="cuda")
        ra = nv.Rows(a16.data_ptr(), 0, lda, 1, M); rc = nv.Rows(c.data_ptr(), 0, N, 1, M)
        f = lambda: nv.lib.lidbox_gemm_bf16s_nt(ra, nv.ptr(b16), ldb, rc, None, K, N, 0, None, None, 0, st)
        nv.check(f())
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): f()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 5 * 1e3)
        us = statistics.median(ts)
        print("lda %5d ldb %5d  %7.1f us  %6.1f TF/s" % (lda, ldb, us, 2.0 * M * N * K / us / 1e6), flush=True)
