"""time vs tile count for one tile shape: the quantisation staircase of the rows kernels (fp32, nn, N=512)"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lidbox_amd import _native as nv
plan = sys.argv[1] if len(sys.argv) > 1 else "64,64,1"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1536
N = 512
os.environ["LIDBOX_GEMM_PLAN"] = plan
os.environ["LIDBOX_GEMM_NO_TAIL_SPLIT"] = "1"
bm = int(plan.split(",")[0]); bn = int(plan.split(",")[1])
st = nv.current_stream()
big = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
Mmax = 64 * 1024
x = torch.randn(Mmax, K, device="cuda"); W = torch.randn(K, N, device="cuda") * 0.05
bias = torch.randn(N, device="cuda"); y = torch.zeros(Mmax, N, device="cuda")
def run(M):
    A = nv.Rows(x.data_ptr(), 0, K, 1, M); Y = nv.Rows(y.data_ptr(), 0, N, 1, M)
    return nv.lib.lidbox_gemm_nn(A, nv.ptr(W), N, Y, K, N, nv.EPI_BIAS_RELU, nv.ptr(bias), nv.ptr(big), big.numel(), st)
def timeit(M, reps=20):
    run(M); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(reps): run(M)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best
tn = N // bn
for tpc in [0.5, 1, 2, 3, 4, 4.125, 4.5, 5, 5.5, 6, 6.25, 7, 8, 9, 10, 12, 12.375, 13, 18, 24, 24.75, 30, 36, 48]:
    tiles = int(tpc * 256)
    M = tiles // tn * bm
    if M > Mmax: break
    t = timeit(M)
    print("plan %s K=%d tiles/CU %6.3f M=%6d  %8.1f us  %6.1f TF" % (plan, K, tiles / 256, M, t, 2.0 * M * N * K / t * 1e-6))
