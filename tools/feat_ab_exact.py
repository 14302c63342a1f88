"""Bit-exact comparison + interleaved timing of feature-kernel library variants on every kind (tools/ab_build.py libraries).
usage: python tools/feat_ab_exact.py ref.so new.so [more.so ...]"""
import ctypes as C
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lidbox_amd import _native as nv
from tools.ab_feat import load

libs = [(os.path.basename(p), load(p)) for p in sys.argv[1:]]
st = nv.current_stream()
KINDS = (("spec", nv.FEAT_SPECTROGRAM, 257), ("mel", nv.FEAT_MEL, 40), ("logmel", nv.FEAT_LOGMEL, 40), ("mfcc", nv.FEAT_MFCC, 12))
plans = {}
for n, lib in libs:
    h = C.c_void_p()
    nv.check(lib.lidbox_feat_plan_create(16000, 400, 160, 512, 2.0, 40, 0.0, 8000.0, 1, 13, C.byref(h)))
    plans[n] = h
torch.manual_seed(0)
for B, N in ((1, 400), (3, 1000), (5, 32001), (7, 4000), (64, 16000), (256, 32000), (2048, 32000)):
    x = torch.randn(B, N, device="cuda") * 0.1
    T = 1 + (N - 400) // 160
    for kn, kind, ch in KINDS:
        ref = None
        for n, lib in libs:
            out = torch.full((B, T, ch), float("nan"), device="cuda")
            nv.check(lib.lidbox_extract_features_fwd(plans[n], kind, nv.ptr(x), B, N, N, nv.ptr(out), 0, None, 0, st))
            torch.cuda.synchronize()
            if ref is None:
                ref = out
            else:
                same = torch.equal(out, ref)
                print("B=%4d N=%5d %-6s %-22s %s  max|d| %.3g" % (B, N, kn, n, "bit-identical" if same else "DIFFERENT", float((out - ref).abs().max())), flush=True)
for B in (256, 512, 1024, 2048):
    x = torch.randn(B, 32000, device="cuda") * 0.1
    for kn, kind, ch in KINDS:
        out = torch.empty(B, 198, ch, device="cuda")
        t = {n: [] for n, _ in libs}
        for _ in range(7):
            for n, lib in libs:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    lib.lidbox_extract_features_fwd(plans[n], kind, nv.ptr(x), B, 32000, 32000, nv.ptr(out), 0, None, 0, st)
                e1.record()
                torch.cuda.synchronize()
                t[n].append(e0.elapsed_time(e1) / 20 * 1e3)
        nbytes = B * (32000 * 4 + 198 * ch * 4)
        print("B=%5d %-6s " % (B, kn) + "   ".join("%s %6.1f us (%4.1f %%)" % (n, statistics.median(t[n]), nbytes / statistics.median(t[n]) / 1e3 / 80) for n, _ in libs), flush=True)
