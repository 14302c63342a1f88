"""Yardstick only (not on the product path): what the vendor fp32 GEMM (torch.mm -> rocBLAS / hipBLASLt) reaches on
the x-vector GEMM shapes at B=256, next to tools/gemm_bench.py's numbers for the hand-written family."""
import torch

def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

torch.backends.cuda.matmul.allow_tf32 = False
shapes = [("frame1", 50688, 200, 512), ("frame2", 25344, 1536, 512), ("frame3", 8448, 1536, 512),
          ("frame4", 8448, 512, 512), ("frame5", 8448, 512, 1500), ("segment1", 256, 3000, 512)]
for name, M, K, N in shapes:
    a = torch.randn(M, K, device="cuda"); w = torch.randn(K, N, device="cuda"); dy = torch.randn(M, N, device="cuda")
    y = torch.empty(M, N, device="cuda"); dx = torch.empty(M, K, device="cuda"); dw = torch.empty(K, N, device="cuda")
    fl = 2.0 * M * K * N
    for tag, fn in (("NN fwd", lambda: torch.mm(a, w, out=y)), ("NT dgrad", lambda: torch.mm(dy, w.t(), out=dx)),
                    ("TN wgrad", lambda: torch.mm(a.t(), dy, out=dw))):
        us = timeit(fn)
        print("%-8s %-8s M=%6d K=%5d N=%5d %8.1f us %6.1f TF/s" % (name, tag, M, K, N, us, fl / us / 1e6))
