"""Pooling kernels in isolation: frame5 output of the x-vector at bs 256 ([256, 33, 1500] fp32, 50.7 MB) and the CNN's
conv_4 output ([256, 99, 3000]).  Algorithmic bytes: forward = read once; backward = read x + write dx."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lidbox_amd import _native as nv


def timeit(fn, reps=30):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


import ctypes as _C
if os.environ.get("LIDBOX_POOL_LIB"):
    _l = _C.CDLL(os.path.abspath(os.environ["LIDBOX_POOL_LIB"]))
    for _n, (_r, _a) in nv._SIGS.items():
        getattr(_l, _n).restype, getattr(_l, _n).argtypes = _r, _a
    nv.lib = _l
st = nv.current_stream()
for B, T, C, stats in ((256, 33, 1500, True), (256, 99, 3000, False), (2048, 33, 1500, True)):
    x = torch.randn(B, T, C, device="cuda")
    out = torch.empty(B, 2 * C if stats else C, device="cuda")
    dout = torch.randn_like(out)
    dx = torch.empty_like(x)
    fwd = nv.lib.lidbox_stats_pool_fwd if stats else nv.lib.lidbox_avg_pool_fwd
    bwd = nv.lib.lidbox_stats_pool_bwd if stats else nv.lib.lidbox_avg_pool_bwd
    nb = x.numel() * 4
    us = timeit(lambda: nv.check(fwd(nv.ptr(x), B, T, C, T * C, C, nv.ptr(out), st)))
    print("[%d,%d,%d] %s fwd %7.1f us  %6.0f GB/s" % (B, T, C, "stats" if stats else "avg", us, nb / us / 1e3))
    if stats:
        us = timeit(lambda: nv.check(bwd(nv.ptr(x), nv.ptr(out), nv.ptr(dout), B, T, C, T * C, C, 1, nv.ptr(dx), st)))
    else:
        us = timeit(lambda: nv.check(bwd(nv.ptr(x), nv.ptr(dout), B, T, C, T * C, C, 1, nv.ptr(dx), st)))
    print("[%d,%d,%d] %s bwd %7.1f us  %6.0f GB/s" % (B, T, C, "stats" if stats else "avg", us, 2 * nb / us / 1e3))
