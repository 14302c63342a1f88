"""conv-size dgrad + wgrad as one launch (queued blocks) vs the two launches: LIDBOX_GEMM_PAIR_MAX_BLOCKS lifted"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lidbox_amd import _native as nv
os.environ["LIDBOX_GEMM_SK"] = "0"
B = 256
st = nv.current_stream()
def R(t, bs, rs, batch, rpb, off=0): return nv.Rows(t.data_ptr() + 4 * off, bs, rs, batch, rpb)
for name, T, Cc, k, s, Co, sp in [("frame3", 99, 512, 3, 3, 512, 8), ("frame4", 33, 512, 1, 1, 512, 24), ("frame5", 33, 512, 1, 1, 1500, 8)]:
    To, Tp = (T - 1) // s + 1, T + k - 1
    x = torch.randn(B, Tp, Cc, device="cuda"); W = torch.randn(k * Cc, Co, device="cuda") * 0.05
    dy = torch.randn(B, To, Co, device="cuda"); dx = torch.zeros(B, Tp, Cc, device="cuda")
    dW = torch.zeros(k * Cc, Co, device="cuda"); db = torch.zeros(Co, device="cuda")
    M, K = B * To, k * Cc
    A, DY = R(x, Tp * Cc, s * Cc, B, To), R(dy, To * Co, Co, B, To)
    nt = min(s, k); Cd = R(dx, Tp * Cc, s * Cc, B, To)
    ws1 = torch.empty(256 << 20, dtype=torch.uint8, device="cuda"); ws2 = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    os.environ["LIDBOX_GEMM_TN_PLAN"] = "64,64,%d" % sp
    os.environ["LIDBOX_GEMM_PLAN"] = "64,64,1"
    def pair():
        nv.check(nv.lib.lidbox_gemm_nt_tn(DY, nv.ptr(W), Co, Cd, Co, nt * Cc, nv.EPI_RELU_MASK, nv.ptr(x), nv.ptr(ws1), ws1.numel(),
                                          A, nv.ptr(dW), Co, K, 0, nv.ptr(db), nv.ptr(ws2), ws2.numel(), st))
    def time(reps=20):
        best = 1e9
        for _ in range(5):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps): pair()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / reps * 1e3)
        return best
    os.environ["LIDBOX_GEMM_PAIR_MAX_BLOCKS"] = "0"
    t_sep = time()
    os.environ["LIDBOX_GEMM_PAIR_MAX_BLOCKS"] = "100000"
    t_pair = time()
    out = (nv.C.c_int * 3)(); nv.lib.lidbox_gemm_last_launches(out)
    print("%-8s two launches %6.1f us   one launch %6.1f us  (launches %s)" % (name, t_sep, t_pair, list(out)))
