"""fp32-accurate GEMM on the bf16 matrix pipe: every forward / dgrad GEMM shape of configs[1] (x-vector, bs 256)
as a bf16 GEMM over split operands, against the native fp32 kernels -- error vs the float64 product and time per call.

a = a0 + a1 + a2 with a0 = bf16(a), a1 = bf16(a - a0), a2 = bf16(a - a0 - a1): 8 significant bits each, exact for fp32 (24).
C = sum_k a_k b_k keeps the products of order >= 2^-16 (6 of the 9): a0 b0, a0 b1, a1 b0, a1 b1, a0 b2, a2 b0; with 9 nothing is
dropped.  As ONE bf16 GEMM over a K' = 6 K (9 K) contraction: A' = [a2 | a0 | a1 | a1 | a0 | a0], B' = [b0 | b2 | b1 | b0 | b1 | b0]
(small terms first), fp32 accumulate in the MFMA -- lidbox_gemm_bf16s_nt as it is (the 256 x 256 ping-pong tile or the LDS-DMA
tiles, whatever the policy picks).  The concatenated operands are built here with torch (tool only); a product kernel would read
three planes per operand and walk the plane pairs in its K loop (same LDS / MFMA work per pair, half the HBM bytes of A').

usage: python tools/split_gemm_experiment.py [B=256]  ->  table; copy to profiles/r05_split_bf16x6_gemm.txt"""
import os
import statistics
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lidbox_amd import _native as nv

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
PAIRS6 = [(2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)]                    # (plane of a, plane of b), small products first
PAIRS9 = [(2, 2), (2, 1), (1, 2)] + PAIRS6


def split3(x):
    p0 = x.bfloat16()
    r1 = x - p0.float()
    p1 = r1.bfloat16()
    p2 = (r1 - p1.float()).bfloat16()
    return [p0, p1, p2]


def timeit(fn, reps=10, rounds=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return statistics.median(ts)


def main():
    st = nv.current_stream()
    # (name, M, K, N): C[M,N] = A[M,K] . B[N,K]^T -- forward (B = W^T) and dgrad (B = W) of the conv layers, SURVEY 8a
    shapes = [("frame1 fwd", 198 * B, 200, 512), ("frame2 fwd", 99 * B, 1536, 512), ("frame3 fwd", 33 * B, 1536, 512),
              ("frame4 fwd", 33 * B, 512, 512), ("frame5 fwd", 33 * B, 512, 1504),
              ("frame5 dgrad", 33 * B, 1504, 512), ("frame4 dgrad", 33 * B, 512, 512), ("frame3 dgrad", 33 * B, 512, 1536),
              ("frame2 dgrad", 99 * B, 512, 1536)]
    print("%-14s %7s %5s %5s | %9s %9s | %9s %9s %6s | %9s %9s %6s" % ("GEMM", "M", "K", "N", "fp32 us", "fp32 err", "x6 us", "x6 err", "x6/f32",
                                                                    "x9 us", "x9 err", "x9/f32"))
    tot = dict(f32=0.0, x6=0.0, x9=0.0)
    torch.manual_seed(0)
    for name, M, K, N in shapes:
        a = torch.randn(M, K, device="cuda")
        b = torch.randn(N, K, device="cuda") * 0.05
        ref = (a.double() @ b.double().T)
        scale = float(ref.abs().max())
        ws = torch.empty(max(16, nv.lib.lidbox_gemm_rows_workspace(M, N, K)), dtype=torch.uint8, device="cuda")
        c = torch.empty(M, N, device="cuda")
        ra, rc = nv.Rows(a.data_ptr(), 0, K, 1, M), nv.Rows(c.data_ptr(), 0, N, 1, M)
        f32 = lambda: nv.check(nv.lib.lidbox_gemm_nt(ra, nv.ptr(b), K, rc, K, N, nv.EPI_NONE, None, nv.ptr(ws), ws.numel(), st))
        t32 = timeit(f32)
        e32 = float((c.double() - ref).abs().max()) / scale
        ap, bp = split3(a), split3(b)
        res = {}
        for tag, pairs in (("x6", PAIRS6), ("x9", PAIRS9)):
            A2 = torch.cat([ap[i] for i, _ in pairs], dim=1).contiguous()
            B2 = torch.cat([bp[j] for _, j in pairs], dim=1).contiguous()
            K2 = A2.shape[1]
            ws2 = torch.empty(max(16, nv.lib.lidbox_gemm_bf16_rows_workspace(M, N, K2)), dtype=torch.uint8, device="cuda")
            c2 = torch.empty(M, N, device="cuda")
            r2, rc2 = nv.Rows(A2.data_ptr(), 0, K2, 1, M), nv.Rows(c2.data_ptr(), 0, N, 1, M)
            fn = lambda: nv.check(nv.lib.lidbox_gemm_bf16s_nt(r2, nv.ptr(B2), K2, rc2, None, K2, N, nv.EPI_NONE, None, nv.ptr(ws2), ws2.numel(), st))
            t = timeit(fn)
            res[tag] = (t, float((c2.double() - ref).abs().max()) / scale)
            del A2, B2, c2, ws2
        print("%-14s %7d %5d %5d | %9.1f %9.2e | %9.1f %9.2e %6.2f | %9.1f %9.2e %6.2f" % (
            name, M, K, N, t32, e32, res["x6"][0], res["x6"][1], res["x6"][0] / t32, res["x9"][0], res["x9"][1], res["x9"][0] / t32), flush=True)
        tot["f32"] += t32; tot["x6"] += res["x6"][0]; tot["x9"] += res["x9"][0]
        del a, b, c, ref, ap, bp
    print("%-14s %19s | %9.1f %9s | %9.1f %9s %6.2f | %9.1f %9s %6.2f" % ("sum", "", tot["f32"], "", tot["x6"], "", tot["x6"] / tot["f32"],
                                                                   tot["x9"], "", tot["x9"] / tot["f32"]))
    print("err = max |C - float64 product| / max |product|; us = median of 5 x 10 back-to-back calls (fp32: lidbox_gemm_nt, native "
          "v_mfma_f32_32x32x2_f32; x6 / x9: lidbox_gemm_bf16s_nt over the concatenated planes, fp32 output)")


if __name__ == "__main__":
    main()
