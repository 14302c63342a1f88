"""Timing only (no parity check: ablation builds compute garbage) of feature-kernel library variants, interleaved in one process.
usage: python tools/feat_time.py [--kind mfcc] lib.so[:NW=n] ...      (NW -> LIDBOX_FEAT_STREAM_NW for that variant's launches)"""
import ctypes as C
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lidbox_amd import _native as nv
from tools.ab_feat import load

argv = sys.argv[1:]
kind, ch = nv.FEAT_LOGMEL, 40
if argv and argv[0] == "--kind":
    kind, ch = {"mfcc": (nv.FEAT_MFCC, 12), "spec": (nv.FEAT_SPECTROGRAM, 257), "mel": (nv.FEAT_MEL, 40), "logmel": (nv.FEAT_LOGMEL, 40)}[argv[1]]
    argv = argv[2:]
libs = []
cache = {}
for spec in argv:
    path, _, opt = spec.partition(":")
    if path not in cache:
        cache[path] = load(path)
    libs.append((os.path.basename(path)[3:-3] + (":" + opt if opt else ""), cache[path], opt.split("=")[1] if opt.startswith("NW=") else None))
st = nv.current_stream()
for B in (256, 2048):
    xs = [torch.randn(B, 32000, device="cuda") * 0.1 for _ in range(1 if B > 256 else 12)]     # B = 256: also a rotation over 12 buffers (cold reads)
    out = torch.empty(B, 198, ch, device="cuda")
    plans = {}
    for n, lib, _ in libs:
        h = C.c_void_p()
        nv.check(lib.lidbox_feat_plan_create(16000, 400, 160, 512, 2.0, 40, 0.0, 8000.0, 1, 13, C.byref(h)))
        plans[n] = h
    for rot in ((False, True) if B == 256 else (False,)):
        t = {n: [] for n, _, _ in libs}
        for _ in range(7):
            for n, lib, nw in libs:
                if nw is None:
                    os.environ.pop("LIDBOX_FEAT_STREAM_NW", None)
                else:
                    os.environ["LIDBOX_FEAT_STREAM_NW"] = nw
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(24):
                    x = xs[i % len(xs)] if rot else xs[0]
                    lib.lidbox_extract_features_fwd(plans[n], kind, nv.ptr(x), B, 32000, 32000, nv.ptr(out), 0, None, 0, st)
                e1.record()
                torch.cuda.synchronize()
                t[n].append(e0.elapsed_time(e1) / 24 * 1e3)
        for n, _, _ in libs:
            print("B=%5d %s %-24s %7.1f us" % (B, "rotating inputs" if rot else "one input      ", n, statistics.median(t[n])), flush=True)
os.environ.pop("LIDBOX_FEAT_STREAM_NW", None)
