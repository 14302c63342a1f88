"""Same-process A/B timing of feature-kernel library variants (tools/ab_build.py).
usage: python tools/ab_feat.py [--kind logmel|mfcc] lib1.so lib2.so[:ITERS=n] ...
A variant may carry LIDBOX_FEAT_ITERS for its launches (name.so:ITERS=2)."""
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
        try:
            fn = getattr(lib, name)
        except AttributeError:
            continue                      # older revisions lack newer entry points
        fn.restype, fn.argtypes = res, args
    return lib


def main():
    argv = sys.argv[1:]
    kind, ch = nv.FEAT_LOGMEL, 40
    if argv and argv[0] == "--kind":
        if argv[1] == "mfcc":
            kind, ch = nv.FEAT_MFCC, 12
        argv = argv[2:]
    libs = []
    for spec in argv:
        path, _, opt = spec.partition(":")
        iters = opt.split("=")[1] if opt.startswith("ITERS=") else None
        libs.append((os.path.basename(path) + (":" + opt if opt else ""), load(path), iters))
    st = nv.current_stream()

    def setenv(iters):
        if iters is None:
            os.environ.pop("LIDBOX_FEAT_ITERS", None)
        else:
            os.environ["LIDBOX_FEAT_ITERS"] = iters

    for B in (256, 2048):
        x = torch.randn(B, 32000, device="cuda") * 0.1
        out = torch.empty(B, 198, ch, device="cuda")
        plans = {}
        for n, lib, _ in libs:
            h = C.c_void_p()
            nv.check(lib.lidbox_feat_plan_create(16000, 400, 160, 512, 2.0, 40, 0.0, 8000.0, 1, 13, C.byref(h)))
            plans[n] = h
        t = {n: [] for n, _, _ in libs}
        ref = None
        for n, lib, iters in libs:
            setenv(iters)
            out.fill_(float("nan"))
            nv.check(lib.lidbox_extract_features_fwd(plans[n], kind, nv.ptr(x), B, 32000, 32000, nv.ptr(out), 0, None, 0, st))
            torch.cuda.synchronize()
            if ref is None:
                ref = out.clone()
            err = float((out - ref).abs().max())
            assert err < 2e-4, (n, err)
        for _ in range(9):
            for n, lib, iters in libs:
                setenv(iters)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    lib.lidbox_extract_features_fwd(plans[n], kind, nv.ptr(x), B, 32000, 32000, nv.ptr(out), 0, None, 0, st)
                e1.record()
                torch.cuda.synchronize()
                t[n].append(e0.elapsed_time(e1) / 10 * 1e3)
        nbytes = B * (32000 * 4 + 198 * ch * 4)
        for n, _, _ in libs:
            us = statistics.median(t[n])
            print("B=%5d %-28s %8.1f us  %7.1f GB/s (%4.1f %% of 8 TB/s)" % (B, n, us, nbytes / us / 1e3, nbytes / us / 1e3 / 80), flush=True)
    setenv(None)


if __name__ == "__main__":
    main()
