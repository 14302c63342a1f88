"""wgrad shapes of the x-vector at bs 256 (and config 5's bs 512): fp32-source bf16 kernel vs the bf16-storage kernel.
usage: python tools/bench_bf16s_tn.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lidbox_amd import _native as nv
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
shapes = [("frame1", B * 198, 200, 512), ("frame2", B * 99, 1536, 512), ("frame3", B * 33, 1536, 512), ("frame4", B * 33, 512, 512),
          ("frame5", B * 33, 512, 1504)]
st = nv.current_stream()
def timeit(fn, n=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
tot = [0.0, 0.0]
for name, M, K1, N in shapes:
    a32, b32 = torch.randn(M, K1, device="cuda"), torch.randn(M, N, device="cuda")
    a16, b16 = a32.bfloat16(), b32.bfloat16()
    c, bg = torch.empty(K1, N, device="cuda"), torch.empty(N, device="cuda")
    w1 = nv.lib.lidbox_gemm_bf16_tn_workspace(M, K1, N); ws1 = torch.empty(w1, dtype=torch.uint8, device="cuda")
    w2 = nv.lib.lidbox_gemm_bf16s_tn_workspace(M, K1, N); ws2 = torch.empty(w2, dtype=torch.uint8, device="cuda")
    old = lambda: nv.check(nv.lib.lidbox_gemm_bf16_tn(nv.Rows(a32.data_ptr(), 0, K1, 1, M), nv.Rows(b32.data_ptr(), 0, N, 1, M), nv.ptr(c), N, K1, N, 0, nv.ptr(bg), nv.ptr(ws1), w1, st))
    new = lambda: nv.check(nv.lib.lidbox_gemm_bf16s_tn(nv.Rows(a16.data_ptr(), 0, K1, 1, M), nv.Rows(b16.data_ptr(), 0, N, 1, M), nv.ptr(c), N, K1, N, 0, nv.ptr(bg), nv.ptr(ws2), w2, st))
    t0, t1 = timeit(old), timeit(new)
    fl = 2.0 * M * K1 * N
    tot[0] += t0; tot[1] += t1
    print("%-7s M=%6d K1=%5d N=%5d  old %7.1f us %6.1f TF   new %7.1f us %6.1f TF" % (name, M, K1, N, t0, fl / t0 * 1e-6, t1, fl / t1 * 1e-6), flush=True)
print("total old %.1f us  new %.1f us (kernel + split reduce)" % tuple(tot))
