"""Interleaved timing of the bf16-storage GEMM (lidbox_gemm_bf16s_nt) against the fp32-source bf16 kernels on the x-vector
layer shapes.  usage: python tools/bench_bf16s.py [utterances per GPU = 256]"""
import os
import statistics
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lidbox_amd import _native as nv

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
SHAPES = [("frame1 fwd", 198 * B, 200, 512), ("frame2 fwd", 99 * B, 1536, 512), ("frame2 dgrad0", 99 * B, 512, 1024),
          ("frame2 dgrad1", 99 * B, 512, 512), ("frame3 fwd", 33 * B, 1536, 512), ("frame3 dgrad", 33 * B, 512, 1536),
          ("frame4 fwd", 33 * B, 512, 512), ("frame5 fwd", 33 * B, 512, 1500)]
st = nv.current_stream()
tot_old = tot_new = 0.0
for name, M, K, N in SHAPES:
    a32 = torch.randn(M, K, device="cuda")
    b32 = torch.randn(N, K, device="cuda")
    a16, b16 = a32.bfloat16(), b32.bfloat16()
    c = torch.empty(M, N, device="cuda")
    c16 = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    wsb = max(16, nv.lib.lidbox_gemm_bf16_rows_workspace(M, N, K))
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    ra32, ra16 = nv.Rows(a32.data_ptr(), 0, K, 1, M), nv.Rows(a16.data_ptr(), 0, K, 1, M)
    rc = nv.Rows(c.data_ptr(), 0, N, 1, M)
    calls = {
        "old": lambda: nv.lib.lidbox_gemm_bf16_nt(ra32, nv.ptr(b32), K, rc, K, N, nv.EPI_BIAS_RELU, nv.ptr(c[0]), nv.ptr(ws), wsb, st),
        "new": lambda: nv.lib.lidbox_gemm_bf16s_nt(ra16, nv.ptr(b16), K, rc, None, K, N, nv.EPI_BIAS_RELU, nv.ptr(c[0]), nv.ptr(ws), wsb, st),
        "new+shadow": lambda: nv.lib.lidbox_gemm_bf16s_nt(ra16, nv.ptr(b16), K, rc, nv.ptr(c16), K, N, nv.EPI_BIAS_RELU, nv.ptr(c[0]), nv.ptr(ws), wsb, st),
    }
    t = {k: [] for k in calls}
    for k, f in calls.items():
        nv.check(f())
    for _ in range(7):
        for k, f in calls.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                f()
            e1.record()
            torch.cuda.synchronize()
            t[k].append(e0.elapsed_time(e1) / 5 * 1e3)
    med = {k: statistics.median(v) for k, v in t.items()}
    fl = 2.0 * M * K * N
    print("%-14s M=%6d K=%5d N=%5d  " % (name, M, K, N) + "  ".join("%s %7.1f us %6.1f TF" % (k, med[k], fl / med[k] / 1e6) for k in calls), flush=True)
    tot_old += med["old"]; tot_new += med["new+shadow"]
print("total old %.1f us  new+shadow %.1f us" % (tot_old, tot_new))
