"""Per-layer GEMM microbenchmark (x-vector shapes at B=256): TFLOP/s of every forward / dgrad / wgrad launch."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lidbox_amd import _native as nv

BF16 = "--bf16" in sys.argv
_pos = [a for a in sys.argv[1:] if not a.startswith("--")]
B = int(_pos[0]) if _pos else 256
REPS = 20
_P = "lidbox_gemm_bf16_" if BF16 else "lidbox_gemm_"
GEMM_NN, GEMM_NT, GEMM_TN = (getattr(nv.lib, _P + n) for n in ("nn", "nt", "tn"))
TN_WORKSPACE = getattr(nv.lib, _P + "tn_workspace")


def rows(t, bs, rs, batch, rpb, off=0):
    return nv.Rows(t.data_ptr() + 4 * off, bs, rs, batch, rpb)


def timeit(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REPS * 1e3   # us


def main():
    st = nv.current_stream()
    total_us, total_fl = 0.0, 0.0
    rws = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    RW, RN = nv.ptr(rws), rws.numel()
    # (name, T_in, C_in, k, s, C_out)
    layers = [("frame1", 198, 40, 5, 1, 512), ("frame2", 198, 512, 3, 2, 512), ("frame3", 99, 512, 3, 3, 512),
              ("frame4", 33, 512, 1, 1, 512), ("frame5", 33, 512, 1, 1, 1500)]
    for name, T, C, k, s, Co in layers:
        To = (T - 1) // s + 1
        Tp = T + k - 1
        x = torch.randn(B, Tp, C, device="cuda")
        W = torch.randn(k * C, Co, device="cuda") * 0.05
        bias = torch.randn(Co, device="cuda")
        y = torch.zeros(B, To, Co, device="cuda")
        dy = torch.randn(B, To, Co, device="cuda")
        dx = torch.zeros(B, Tp, C, device="cuda")
        dW = torch.zeros(k * C, Co, device="cuda")
        M, K = B * To, k * C
        A = rows(x, Tp * C, s * C, B, To)
        Y = rows(y, To * Co, Co, B, To)
        DY = rows(dy, To * Co, Co, B, To)
        fl = 2.0 * M * K * Co
        us = timeit(lambda: nv.check(GEMM_NN(A, nv.ptr(W), Co, Y, K, Co, nv.EPI_BIAS_RELU, nv.ptr(bias), RW, RN, st)))
        print("%-8s fwd   M=%6d K=%5d N=%5d  %8.1f us  %6.1f TF/s" % (name, M, K, Co, us, fl / us / 1e6))
        total_us += us; total_fl += fl
        wsb = TN_WORKSPACE(M, K, Co)
        ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
        us = timeit(lambda: nv.check(GEMM_TN(A, DY, nv.ptr(dW), Co, K, Co, 0, nv.ptr(bias), nv.ptr(ws), wsb, st)))
        print("%-8s wgrad M=%6d K1=%4d N=%5d  %8.1f us  %6.1f TF/s" % (name, M, K, Co, us, fl / us / 1e6))
        total_us += us; total_fl += fl
        if name != "frame1":
            ng = (k + s - 1) // s
            for g in range(ng):
                nt = min(s, k - g * s)
                Cd = rows(dx, Tp * C, s * C, B, To, off=g * s * C)
                Wg = nv.C.c_void_p(W.data_ptr() + 4 * g * s * C * Co)
                mask = nv.C.c_void_p(x.data_ptr() + 4 * g * s * C)
                epi = nv.EPI_RELU_MASK if g == 0 else nv.EPI_ACCUM_RELU_MASK
                flg = 2.0 * M * Co * nt * C
                us = timeit(lambda: nv.check(GEMM_NT(DY, Wg, Co, Cd, Co, nt * C, epi, mask, RW, RN, st)))
                print("%-8s dgrad%d M=%6d K=%5d N=%5d  %8.1f us  %6.1f TF/s" % (name, g, M, Co, nt * C, us, flg / us / 1e6))
                total_us += us; total_fl += flg
    for name, K, N in [("segment1", 3000, 512), ("segment2", 512, 512), ("outputs", 512, 4)]:
        x = torch.randn(B, K, device="cuda")
        W = torch.randn(K, N, device="cuda") * 0.05
        bias = torch.randn(N, device="cuda")
        y = torch.zeros(B, N, device="cuda")
        dy = torch.randn(B, N, device="cuda")
        dx = torch.zeros(B, K, device="cuda")
        dW = torch.zeros(K, N, device="cuda")
        fl = 2.0 * B * K * N
        A, Y, DY, DX = rows(x, 0, K, 1, B), rows(y, 0, N, 1, B), rows(dy, 0, N, 1, B), rows(dx, 0, K, 1, B)
        us = timeit(lambda: nv.check(GEMM_NN(A, nv.ptr(W), N, Y, K, N, nv.EPI_BIAS_RELU, nv.ptr(bias), RW, RN, st)))
        print("%-8s fwd   M=%6d K=%5d N=%5d  %8.1f us  %6.1f TF/s" % (name, B, K, N, us, fl / us / 1e6))
        total_us += us; total_fl += fl
        wsb = TN_WORKSPACE(B, K, N)
        ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device="cuda")
        us = timeit(lambda: nv.check(GEMM_TN(A, DY, nv.ptr(dW), N, K, N, 0, nv.ptr(bias), nv.ptr(ws), wsb, st)))
        print("%-8s wgrad M=%6d K1=%4d N=%5d  %8.1f us  %6.1f TF/s" % (name, B, K, N, us, fl / us / 1e6))
        total_us += us; total_fl += fl
        us = timeit(lambda: nv.check(GEMM_NT(DY, nv.ptr(W), N, DX, N, K, nv.EPI_RELU_MASK, nv.ptr(x), RW, RN, st)))
        print("%-8s dgrad M=%6d K=%5d N=%5d  %8.1f us  %6.1f TF/s" % (name, B, N, K, us, fl / us / 1e6))
        total_us += us; total_fl += fl
    print("TOTAL %.1f us  %.1f GFLOP  %.1f TF/s  -> %.0f utt/s GEMM-only" % (total_us, total_fl / 1e9, total_fl / total_us / 1e6,
                                                                               B / (total_us * 1e-6)))


if __name__ == "__main__":
    main()
