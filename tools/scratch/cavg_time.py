import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from lidbox_amd import _native as nv
st = nv.current_stream()
for B, N, Th in ((512, 100, 100), (64, 100, 100), (2048, 100, 100), (512, 10, 100), (512, 100, 64), (512, 100, 10), (512, 50, 100)):
    sc = -torch.rand(B, N, device="cuda") * 3.14
    y = torch.randint(0, N, (B,), dtype=torch.int32, device="cuda")
    thr = torch.linspace(-3.14159, 0, Th, device="cuda")
    tp, fn = torch.zeros(N, Th, device="cuda"), torch.zeros(N, Th, device="cuda")
    fp, tn = torch.zeros(N, N, Th, device="cuda"), torch.zeros(N, N, Th, device="cuda")
    f = lambda: nv.check(nv.lib.lidbox_cavg_update(nv.ptr(sc), nv.ptr(y), B, N, nv.ptr(thr), Th, nv.ptr(tp), nv.ptr(fn), nv.ptr(fp), nv.ptr(tn), st))
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(9):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 100)
    print("cavg_update B %5d N %4d Th %4d: %.1f us per launch (10 back to back)" % (B, N, Th, statistics.median(ts)))
