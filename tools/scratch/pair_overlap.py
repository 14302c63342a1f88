"""would running a layer's wgrad beside its dgrad (one launch / two streams) save time?  Eager two-stream experiment:
sequential on one stream vs concurrent on two, per x-vector layer at bs 256 (fp32)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lidbox_amd import _native as nv
B = 256
layers = [("frame2", 198, 512, 3, 2, 512), ("frame3", 99, 512, 3, 3, 512), ("frame4", 33, 512, 1, 1, 512), ("frame5", 33, 512, 1, 1, 1500)]
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def R(t, bs, rs, batch, rpb, off=0): return nv.Rows(t.data_ptr() + 4 * off, bs, rs, batch, rpb)
for name, T, Cc, k, s, Co in layers:
    To, Tp = (T - 1) // s + 1, T + k - 1
    x = torch.randn(B, Tp, Cc, device="cuda"); W = torch.randn(k * Cc, Co, device="cuda") * 0.05
    bias = torch.randn(Co, device="cuda"); dy = torch.randn(B, To, Co, device="cuda"); dx = torch.zeros(B, Tp, Cc, device="cuda")
    dW = torch.zeros(k * Cc, Co, device="cuda"); db = torch.zeros(Co, device="cuda")
    M, K = B * To, k * Cc
    A, DY = R(x, Tp * Cc, s * Cc, B, To), R(dy, To * Co, Co, B, To)
    nt = min(s, k)
    Cd = R(dx, Tp * Cc, s * Cc, B, To)
    ws1 = torch.empty(256 << 20, dtype=torch.uint8, device="cuda"); ws2 = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    def dgrad(st, ws): nv.check(nv.lib.lidbox_gemm_nt(DY, nv.ptr(W), Co, Cd, Co, nt * Cc, nv.EPI_RELU_MASK, nv.ptr(x), nv.ptr(ws), ws.numel(), st))
    def wgrad(st, ws): nv.check(nv.lib.lidbox_gemm_tn(A, DY, nv.ptr(dW), Co, K, Co, 0, nv.ptr(db), nv.ptr(ws), ws.numel(), st))
    h1, h2 = nv.C.c_void_p(s1.cuda_stream), nv.C.c_void_p(s2.cuda_stream)
    def seq():
        dgrad(h1, ws1); wgrad(h1, ws1)
    def par():
        s2.wait_stream(s1)
        dgrad(h1, ws1); wgrad(h2, ws2)
        s1.wait_stream(s2)
    def time(fn, reps=20):
        best = 1e9
        for _ in range(5):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(s1):
                e0.record(s1)
                for _ in range(reps): fn()
                e1.record(s1)
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / reps * 1e3)
        return best
    def only(f, ws):
        return time(lambda: f(h1, ws))
    td, tw = only(dgrad, ws1), only(wgrad, ws1)
    print("%-8s dgrad %6.1f  wgrad(+reduce) %6.1f  sequential %6.1f  two streams %6.1f us" % (name, td, tw, time(seq), time(par)))
