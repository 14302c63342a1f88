"""NN vs NT vs TN at identical (M, K, N): isolates the operand-staging cost from shape effects.
usage: python tools/gemm_shapes.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lidbox_amd import _native as nv

REPS = 20


def rows(t, rs, n):
    return nv.Rows(t.data_ptr(), 0, rs, 1, n)


def timeit(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REPS * 1e3


def main():
    st = nv.current_stream()
    rws = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    RW, RN = nv.ptr(rws), rws.numel()
    plan = (nv.C.c_int * 4)()
    for M, K, N in [(25344, 1536, 512), (25344, 512, 1024), (25344, 512, 512), (50688, 200, 512), (8448, 1536, 512),
                    (8448, 512, 512), (8448, 512, 1500), (8448, 1500, 512), (8448, 512, 1536), (256, 3000, 512),
                    (256, 512, 3000), (256, 512, 512)]:
        a = torch.randn(M, K, device="cuda")
        bnn = torch.randn(K, N, device="cuda")
        bnt = torch.randn(N, K, device="cuda")
        c = torch.zeros(M, N, device="cuda")
        fl = 2.0 * M * K * N
        A, Cd = rows(a, K, M), rows(c, N, M)
        nv.lib.lidbox_gemm_plan_query(0, M, N, K, RN, plan)
        p = tuple(plan)
        us_nn = timeit(lambda: nv.check(nv.lib.lidbox_gemm_nn(A, nv.ptr(bnn), N, Cd, K, N, nv.EPI_NONE, None, RW, RN, st)))
        us_nt = timeit(lambda: nv.check(nv.lib.lidbox_gemm_nt(A, nv.ptr(bnt), K, Cd, K, N, nv.EPI_NONE, None, RW, RN, st)))
        print("M=%6d K=%5d N=%5d plan %s  NN %7.1f us %6.1f TF/s   NT %7.1f us %6.1f TF/s" %
              (M, K, N, p, us_nn, fl / us_nn / 1e6, us_nt, fl / us_nt / 1e6), flush=True)


if __name__ == "__main__" and "--epilogues" not in sys.argv:
    main()


def epilogues():
    """cost of the backward epilogues at frame2's dgrad shape (dense operands)"""
    st = nv.current_stream()
    rws = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    RW, RN = nv.ptr(rws), rws.numel()
    for M, K, N in [(25344, 512, 1024), (25344, 512, 512), (8448, 512, 1536)]:
        a = torch.randn(M, K, device="cuda")
        b = torch.randn(N, K, device="cuda")
        c = torch.zeros(M, N, device="cuda")
        mask = torch.randn(M, N, device="cuda")
        fl = 2.0 * M * K * N
        A, Cd = rows(a, K, M), rows(c, N, M)
        for name, epi, aux in [("none", nv.EPI_NONE, None), ("relu_mask", nv.EPI_RELU_MASK, nv.ptr(mask)),
                               ("accum", nv.EPI_ACCUM, None), ("accum_relu_mask", nv.EPI_ACCUM_RELU_MASK, nv.ptr(mask))]:
            us = timeit(lambda: nv.check(nv.lib.lidbox_gemm_nt(A, nv.ptr(b), K, Cd, K, N, epi, aux, RW, RN, st)))
            print("NT M=%6d K=%5d N=%5d  %-16s %7.1f us %6.1f TF/s" % (M, K, N, name, us, fl / us / 1e6), flush=True)


if __name__ == "__main__" and "--epilogues" in sys.argv:
    epilogues()
