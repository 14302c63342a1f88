"""lidbox_cavg_update alone: B examples x N classes x Th thresholds, HIP events around a captured graph of 50 launches (median of 7 replays).
LIDBOX_CAVG_NZ=k forces the number of label chunks along grid.z, LIDBOX_CAVG_UNIT=u the examples per unit of the walk.
usage: python tools/cavg_time.py [B N Th]"""
import os, sys, statistics
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidbox_amd.metrics import SparseAverageDetectionCost

B, N, Th = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (512, 50, 100)
g = torch.Generator().manual_seed(0)
s = (-torch.rand(B, N, generator=g) * 3.14).cuda()
y = torch.randint(0, N, (B,), generator=g, dtype=torch.int32).cuda()
for nz, unit in (("1", "16"), ("1", "32"), ("1", "64"), ("2", "16"), ("2", "32"), ("4", "16"), ("4", "32"), ("", "")):
    for k, v in (("LIDBOX_CAVG_NZ", nz), ("LIDBOX_CAVG_UNIT", unit)):
        if v:
            os.environ[k] = v
        else:
            os.environ.pop(k, None)
    m = SparseAverageDetectionCost(N, np.linspace(-np.pi, 0, Th))
    for _ in range(10):
        m._update_sparse(y, s)
    ts = []
    graph = torch.cuda.CUDAGraph()                     # 50 captured launches: the host's ~13 us per ctypes call stays out of it
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        with torch.cuda.graph(graph, stream=st):
            for _ in range(50):
                m._update_sparse(y, s)
        graph.replay()
        for _ in range(7):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(st)
            graph.replay()
            b.record(st)
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3 / 50)
    print("nz %-7s unit %-7s %7.2f us   (C_avg %.6f)" % (nz or "default", unit or "default", statistics.median(ts), float(m.result())))
