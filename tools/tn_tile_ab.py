"""wgrad (lidbox_gemm_tn) timing under forced decompositions (LIDBOX_GEMM_TN_PLAN is read per call).
usage: python tools/tn_tile_ab.py"""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lidbox_amd import _native as nv
st = nv.current_stream()
SHAPES = (("frame2 wgrad", 25344, 1536, 512, ["128,128,16", "256,128,20", "256,128,21", "256,128,22", "256,128,42"]),
          ("frame2 wgrad bs512", 50688, 1536, 512, ["128,128,16", "256,128,21", "256,128,42"]),
          ("frame2 wgrad bs128", 12672, 1536, 512, ["128,128,16", "256,128,21"]),
          ("frame3 wgrad", 8448, 1536, 512, ["128,128,16", "256,128,21"]),
          ("frame5 wgrad", 8448, 512, 1500, ["128,128,16", "256,128,21"]))
for name, M, K1, N, plans in SHAPES:
    a = torch.randn(M, K1, device="cuda"); b = torch.randn(M, N, device="cuda")
    c = torch.empty(K1, N, device="cuda"); bg = torch.empty(N, device="cuda")
    ref = None
    for plan in plans:
        os.environ["LIDBOX_GEMM_TN_PLAN"] = plan
        wsb = nv.lib.lidbox_gemm_tn_workspace(M, K1, N)
        ws = torch.empty(max(16, wsb), dtype=torch.uint8, device="cuda")
        f = lambda: nv.lib.lidbox_gemm_tn(nv.Rows(a.data_ptr(), 0, K1, 1, M), nv.Rows(b.data_ptr(), 0, N, 1, M), nv.ptr(c), N, K1, N, 0, nv.ptr(bg), nv.ptr(ws), ws.numel(), st)
        nv.check(f())
        torch.cuda.synchronize()
        if ref is None:
            ref = c.clone()
        err = float((c - ref).abs().max() / ref.abs().max())
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): f()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 5 * 1e3)
        us = statistics.median(ts)
        print("%-13s plan %-12s %7.1f us %6.1f TF/s  (rel diff vs first plan %.1e)" % (name, plan, us, 2.0 * M * K1 * N / us / 1e6, err), flush=True)
