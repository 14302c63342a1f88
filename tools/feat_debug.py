"""Compare two library variants' feature outputs elementwise (debug aid).  usage: feat_debug.py ref.so new.so [kind]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lidbox_amd import _native as nv
from tools.ab_feat import load
kind = int(sys.argv[3]) if len(sys.argv) > 3 else nv.FEAT_LOGMEL
ch = {0: 257, 1: 40, 2: 40, 3: 12}[kind]
st = nv.current_stream()
for B in (1, 3, 256):
    torch.manual_seed(0)
    x = torch.randn(B, 32000, device="cuda") * 0.1
    outs = []
    for path in sys.argv[1:3]:
        lib = load(path)
        h = C.c_void_p()
        nv.check(lib.lidbox_feat_plan_create(16000, 400, 160, 512, 2.0, 40, 0.0, 8000.0, 1, 13, C.byref(h)))
        out = torch.full((B, 198, ch), float("nan"), device="cuda")
        nv.check(lib.lidbox_extract_features_fwd(h, kind, nv.ptr(x), B, 32000, 32000, nv.ptr(out), 0, None, 0, st))
        torch.cuda.synchronize()
        outs.append(out.cpu())
    a, b = outs
    bad = ~torch.isfinite(b)
    d = (a - b).abs()
    d[bad] = 0
    print("B=%d nonfinite=%d maxdiff(finite)=%.3g" % (B, int(bad.sum()), float(d.max())))
    if bad.any():
        idx = bad.nonzero()
        print(" first bad (b,t,c):", idx[:12].tolist())
        print(" bad frames mod 8:", sorted(set((idx[:, 1] % 8).tolist())), "bad chans:", sorted(set(idx[:, 2].tolist()))[:40])
    big = (d > 1e-3).nonzero()
    if len(big):
        print(" diffs > 1e-3:", len(big), big[:10].tolist(), "chans", sorted(set(big[:, 2].tolist()))[:40], "t%8", sorted(set((big[:, 1] % 8).tolist())))
    if B == 1:
        torch.set_printoptions(precision=4, linewidth=200)
        print("ref t=8..11 c0..5\n", a[0, 8:12, :6])
        print("new t=8..11 c0..5\n", b[0, 8:12, :6])
        print("ref t=0..1 c0..5\n", a[0, 0:2, :6], "\nnew\n", b[0, 0:2, :6])
