"""Yardstick only (not on the product path): what the vendor bf16 GEMM (torch.mm on bfloat16 -> hipBLASLt) reaches on the
x-vector GEMM shapes at B=256 (bf16 in, bf16 out, fp32 accumulate), next to tools/bench_bf16s.py / step_calls.py's numbers
for the hand-written storage family."""
import torch

def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

shapes = [("frame1", 50688, 200, 512), ("frame2", 25344, 1536, 512), ("frame3", 8448, 1536, 512),
          ("frame4", 8448, 512, 512), ("frame5", 8448, 512, 1504)]
bf = torch.bfloat16
for name, M, K, N in shapes:
    a = torch.randn(M, K, device="cuda", dtype=bf); wt = torch.randn(N, K, device="cuda", dtype=bf); dy = torch.randn(M, N, device="cuda", dtype=bf)
    w = wt.t().contiguous()
    y = torch.empty(M, N, device="cuda", dtype=bf); dx = torch.empty(M, K, device="cuda", dtype=bf); dw = torch.empty(K, N, device="cuda", dtype=bf)
    fl = 2.0 * M * K * N
    for tag, fn in (("NT fwd (B k-inner)", lambda: torch.mm(a, wt.t(), out=y)), ("NN fwd", lambda: torch.mm(a, w, out=y)),
                    ("dgrad (dy @ W^T)", lambda: torch.mm(dy, w.t(), out=dx)), ("dgrad k-inner", lambda: torch.mm(dy, wt, out=dx)),
                    ("TN wgrad", lambda: torch.mm(a.t(), dy, out=dw))):
        us = timeit(fn)
        print("%-8s %-20s M=%6d K=%5d N=%5d %8.1f us %7.1f TF/s" % (name, tag, M, K, N, us, fl / us / 1e6), flush=True)
