"""Same-process A/B timing of GEMM library variants built by tools/ab_build.py: every x-vector GEMM launch
(B = 256) is timed alternately on each library, several rounds, medians reported.
usage: python tools/ab_gemm.py tools/ab/libA.so tools/ab/libB.so [...]
A variant is a library path, optionally followed by environment settings that apply to its launches only:
    python tools/ab_gemm.py lidbox_amd/csrc/liblidbox_hip.so:LIDBOX_GEMM_SK=0 lidbox_amd/csrc/liblidbox_hip.so:LIDBOX_GEMM_SK=1"""
import ctypes as C
import os
import statistics
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lidbox_amd import _native as nv

B, REPS, ROUNDS = 256, 10, 7


def load(path):
    lib = C.CDLL(os.path.abspath(path))
    for name, (res, args) in nv._SIGS.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    return lib


def rows(t, bs, rs, batch, rpb, off=0):
    return nv.Rows(t.data_ptr() + 4 * off, bs, rs, batch, rpb)


def timeit(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REPS * 1e3


def main():
    libs, envs = [], {}
    for arg in sys.argv[1:]:
        path, _, env = arg.partition(":")
        name = os.path.basename(path) + (":" + env if env else "")
        libs.append((name, load(path)))
        envs[name] = dict(kv.split("=", 1) for kv in env.split(",")) if env else {}
    all_keys = sorted({k for e in envs.values() for k in e})

    def use(name):
        for k in all_keys:
            os.environ.pop(k, None)
        os.environ.update(envs[name])
    st = nv.current_stream()
    rws = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    RW, RN = nv.ptr(rws), rws.numel()
    cases = []
    layers = [("frame1", 198, 40, 5, 1, 512), ("frame2", 198, 512, 3, 2, 512), ("frame3", 99, 512, 3, 3, 512),
              ("frame4", 33, 512, 1, 1, 512), ("frame5", 33, 512, 1, 1, 1500)]
    keep = []
    for name, T, Cc, k, s, Co in layers:
        To, Tp = (T - 1) // s + 1, T + k - 1
        x = torch.randn(B, Tp, Cc, device="cuda"); W = torch.randn(k * Cc, Co, device="cuda") * 0.05
        bias = torch.randn(Co, device="cuda"); y = torch.zeros(B, To, Co, device="cuda")
        dy = torch.randn(B, To, Co, device="cuda"); dx = torch.zeros(B, Tp, Cc, device="cuda")
        dW = torch.zeros(k * Cc, Co, device="cuda")
        keep += [x, W, bias, y, dy, dx, dW]
        M, K = B * To, k * Cc
        A, Y, DY = rows(x, Tp * Cc, s * Cc, B, To), rows(y, To * Co, Co, B, To), rows(dy, To * Co, Co, B, To)
        cases.append((name + " fwd", 2.0 * M * K * Co,
                      lambda lib, A=A, W=W, Co=Co, Y=Y, K=K, bias=bias: lib.lidbox_gemm_nn(A, nv.ptr(W), Co, Y, K, Co, nv.EPI_BIAS_RELU, nv.ptr(bias), RW, RN, st)))
        wsb = max(lib.lidbox_gemm_tn_workspace(M, K, Co) for _, lib in libs)
        ws = torch.empty(wsb, dtype=torch.uint8, device="cuda"); keep.append(ws)
        cases.append((name + " wgrad", 2.0 * M * K * Co,
                      lambda lib, A=A, DY=DY, dW=dW, Co=Co, K=K, bias=bias, ws=ws, wsb=wsb: lib.lidbox_gemm_tn(A, DY, nv.ptr(dW), Co, K, Co, 0, nv.ptr(bias), nv.ptr(ws), wsb, st)))
        if name != "frame1":
            for g in range((k + s - 1) // s):
                nt = min(s, k - g * s)
                Cd = rows(dx, Tp * Cc, s * Cc, B, To, off=g * s * Cc)
                Wg = C.c_void_p(W.data_ptr() + 4 * g * s * Cc * Co)
                mask = C.c_void_p(x.data_ptr() + 4 * g * s * Cc)
                epi = nv.EPI_RELU_MASK if g == 0 else nv.EPI_ACCUM_RELU_MASK
                cases.append(("%s dgrad%d" % (name, g), 2.0 * M * Co * nt * Cc,
                              lambda lib, DY=DY, Wg=Wg, Co=Co, Cd=Cd, n=nt * Cc, epi=epi, mask=mask: lib.lidbox_gemm_nt(DY, Wg, Co, Cd, Co, n, epi, mask, RW, RN, st)))
    for name, K, N in [("segment1", 3000, 512), ("segment2", 512, 512), ("outputs", 512, 4)]:
        x = torch.randn(B, K, device="cuda"); W = torch.randn(K, N, device="cuda") * 0.05
        bias = torch.randn(N, device="cuda"); y = torch.zeros(B, N, device="cuda")
        dy = torch.randn(B, N, device="cuda"); dx = torch.zeros(B, K, device="cuda"); dW = torch.zeros(K, N, device="cuda")
        keep += [x, W, bias, y, dy, dx, dW]
        A, Y, DY, DX = rows(x, 0, K, 1, B), rows(y, 0, N, 1, B), rows(dy, 0, N, 1, B), rows(dx, 0, K, 1, B)
        fl = 2.0 * B * K * N
        cases.append((name + " fwd", fl, lambda lib, A=A, W=W, N=N, Y=Y, K=K, bias=bias: lib.lidbox_gemm_nn(A, nv.ptr(W), N, Y, K, N, nv.EPI_BIAS_RELU, nv.ptr(bias), RW, RN, st)))
        wsb = max(16, max(lib.lidbox_gemm_tn_workspace(B, K, N) for _, lib in libs))
        ws = torch.empty(wsb, dtype=torch.uint8, device="cuda"); keep.append(ws)
        cases.append((name + " wgrad", fl, lambda lib, A=A, DY=DY, dW=dW, N=N, K=K, bias=bias, ws=ws, wsb=wsb: lib.lidbox_gemm_tn(A, DY, nv.ptr(dW), N, K, N, 0, nv.ptr(bias), nv.ptr(ws), wsb, st)))
        cases.append((name + " dgrad", fl, lambda lib, DY=DY, W=W, N=N, DX=DX, K=K, x=x: lib.lidbox_gemm_nt(DY, nv.ptr(W), N, DX, N, K, nv.EPI_RELU_MASK, nv.ptr(x), RW, RN, st)))
    totals = {n: 0.0 for n, _ in libs}
    print("%-16s" % "launch" + "".join("%22s" % n[-22:] for n, _ in libs))
    for cname, fl, fn in cases:
        t = {n: [] for n, _ in libs}
        for n, lib in libs:
            use(n)
            nv.check(fn(lib))
        torch.cuda.synchronize()
        for _ in range(ROUNDS):
            for n, lib in libs:
                use(n)
                t[n].append(timeit(lambda: fn(lib)))
        line = "%-16s" % cname
        for n, _ in libs:
            med = statistics.median(t[n])
            totals[n] += med
            line += "%12.1f us %5.1f TF" % (med, fl / med / 1e6)
        print(line, flush=True)
    print("%-16s" % "TOTAL" + "".join("%12.1f us         " % totals[n] for n, _ in libs))


if __name__ == "__main__":
    main()
