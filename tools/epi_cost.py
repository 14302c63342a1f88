"""What the ReLU-mask / accumulate epilogues cost on the dgrad shapes (fp32 family, nt): the same launch with a plain store.
usage: python tools/epi_cost.py"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lidbox_amd import _native as nv
st = nv.current_stream()
ws = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
SH = [("frame5 dgrad", 8448, 1500, 512), ("frame4 dgrad", 8448, 512, 512), ("frame3 dgrad", 8448, 512, 1536), ("frame2 dgrad g0", 25344, 512, 1024), ("frame2 dgrad g1", 25344, 512, 512)]
BATCH = 256 if "--batched" in sys.argv else 1        # --batched: C rows as 256 utterances (an utterance boundary every 33 / 99 rows, as in the step)
EPIS = [("none", nv.EPI_NONE), ("relu_mask", nv.EPI_RELU_MASK), ("accum", nv.EPI_ACCUM), ("accum_relu_mask", nv.EPI_ACCUM_RELU_MASK)]
for name, M, K, N in SH:
    x = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda"); y = torch.zeros(M, N, device="cuda"); mask = torch.randn(M, N, device="cuda")
    calls = {}
    for en, e in EPIS:
        calls[en] = (lambda e=e: nv.check(nv.lib.lidbox_gemm_nt(nv.Rows(x.data_ptr(), 0, K, 1, M), nv.ptr(W), K, nv.Rows(y.data_ptr(), (M // BATCH) * N, N, BATCH, M // BATCH), K, N, e,
                                                                nv.ptr(mask) if e in (nv.EPI_RELU_MASK, nv.EPI_ACCUM_RELU_MASK) else None, nv.ptr(ws), ws.numel(), st)))
    t = {k: [] for k in calls}
    for f in calls.values():
        f()
    for _ in range(7):
        for k, f in calls.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                f()
            e1.record(); torch.cuda.synchronize()
            t[k].append(e0.elapsed_time(e1) / 5 * 1e3)
    fl = 2.0 * M * K * N
    print("%-16s M=%6d K=%5d N=%5d  " % (name, M, K, N) + "  ".join("%s %6.1f us %5.1f TF" % (k, statistics.median(v), fl / statistics.median(v) / 1e6) for k, v in t.items()), flush=True)
