"""decomposition sweep of ONE fp32 NT launch with the ReLU-mask epilogue (LIDBOX_GEMM_PLAN / _NO_TAIL_SPLIT overrides),
interleaved medians.  usage: python tools/sweep_one_nt.py M K N [A row stride]"""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lidbox_amd import _native as nv
M, K, N = (int(v) for v in sys.argv[1:4])
rs = int(sys.argv[4]) if len(sys.argv) > 4 else K
a = torch.randn(M * rs + K, device="cuda"); b = torch.randn(N, K, device="cuda"); c = torch.empty(M, N, device="cuda")
mask = torch.randn(M, N, device="cuda")
ws = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
st = nv.current_stream()
call = lambda: nv.check(nv.lib.lidbox_gemm_nt(nv.Rows(a.data_ptr(), 0, rs, 1, M), nv.ptr(b), K, nv.Rows(c.data_ptr(), 0, N, 1, M), K, N,
                                              nv.EPI_RELU_MASK, nv.ptr(mask), nv.ptr(ws), ws.numel(), st))
cands = [("model", None, None)] + [("%dx%d nts=%d" % (bm, bn, nts), "%d,%d,1" % (bm, bn), nts) for bm, bn in ((64, 64), (128, 64), (64, 128), (128, 128)) for nts in (0, 1)]
times = {n: [] for n, _, _ in cands}
for rep in range(7):
    for name, plan, nts in cands:
        for k in ("LIDBOX_GEMM_PLAN", "LIDBOX_GEMM_NO_TAIL_SPLIT"):
            os.environ.pop(k, None)
        if plan:
            os.environ["LIDBOX_GEMM_PLAN"] = plan
            if nts: os.environ["LIDBOX_GEMM_NO_TAIL_SPLIT"] = "1"
        call(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): call()
        e1.record(); torch.cuda.synchronize()
        times[name].append(e0.elapsed_time(e1) / 5 * 1e3)
for name, _, _ in cands:
    us = statistics.median(times[name])
    print("%-16s %7.1f us  %6.1f TF" % (name, us, 2.0 * M * K * N / us * 1e-6))
