"""Same-process A/B timing of feature-kernel library variants (tools/ab_build.py).
usage: python tools/ab_feat.py lib1.so lib2.so ..."""
import ctypes as C
import os
import statistics
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lidbox_amd import _native as nv


def load(path):
    lib = C.CDLL(os.path.abspath(path))
    for name, (res, args) in nv._SIGS.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    return lib


def main():
    libs = [(os.path.basename(p), load(p)) for p in sys.argv[1:]]
    st = nv.current_stream()
    for B in (256, 2048):
        x = torch.randn(B, 32000, device="cuda") * 0.1
        out = torch.empty(B, 198, 40, device="cuda")
        plans = {}
        for n, lib in libs:
            h = C.c_void_p()
            nv.check(lib.lidbox_feat_plan_create(16000, 400, 160, 512, 2.0, 40, 0.0, 8000.0, 1, 13, C.byref(h)))
            plans[n] = h
        t = {n: [] for n, _ in libs}
        ref = None
        for n, lib in libs:
            nv.check(lib.lidbox_extract_features_fwd(plans[n], nv.FEAT_LOGMEL, nv.ptr(x), B, 32000, 32000, nv.ptr(out), 0, None, 0, st))
            torch.cuda.synchronize()
            if ref is None:
                ref = out.clone()
            assert float((out - ref).abs().max()) < 1e-4, (n, float((out - ref).abs().max()))
        for _ in range(9):
            for n, lib in libs:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    lib.lidbox_extract_features_fwd(plans[n], nv.FEAT_LOGMEL, nv.ptr(x), B, 32000, 32000, nv.ptr(out), 0, None, 0, st)
                e1.record()
                torch.cuda.synchronize()
                t[n].append(e0.elapsed_time(e1) / 10 * 1e3)
        for n, _ in libs:
            us = statistics.median(t[n])
            print("B=%5d %-22s %8.1f us  %7.1f GB/s (%4.1f %% of 8 TB/s)" % (B, n, us, B * 159680 / us / 1e3, B * 159680 / us / 1e3 / 80))


if __name__ == "__main__":
    main()
