"""Debug: whole-kernel timeline of the fused feature kernel per wave (entry, tables staged, the k-th tile's start / end or phases, exit).
Libraries: `LIDBOX_AB_DIR=ab_ship python tools/ab_build.py feattl0 features.hip -DLBX_FEAT_TIMELINE=0` (k-th tile = 0; =1 second tile;
-DLBX_FEAT_TIMING=k adds the fenced per-phase stamps).  Run: `python tools/feat_timeline.py tools/ab_ship/libfeattl0.so [B] [kind]`.
s_memtime is per XCD (different bases): waves are placed on the common axis through s_memrealtime (100 MHz) at entry."""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lidbox_amd import _native as nv
from tools.ab_feat import load

lib = load(sys.argv[1])
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
kind = {"logmel": nv.FEAT_LOGMEL, "mfcc": nv.FEAT_MFCC}[sys.argv[3] if len(sys.argv) > 3 else "logmel"]
ch = 40 if kind == nv.FEAT_LOGMEL else 12
x = torch.randn(B, 32000, device="cuda") * 0.1
h = C.c_void_p()
nv.check(lib.lidbox_feat_plan_create(16000, 400, 160, 512, 2.0, 40, 0.0, 8000.0, 1, 13, C.byref(h)))
out = torch.empty(B, 198, ch, device="cuda")
stamps = torch.zeros(4096 * 4 * 16, dtype=torch.int64, device="cuda")
for _ in range(4):
    stamps.zero_()
    torch.cuda.synchronize()
    nv.check(lib.lidbox_extract_features_fwd(h, kind, nv.ptr(x), B, 32000, 32000, nv.ptr(out), 0, nv.ptr(stamps), stamps.numel() * 8,
                                             nv.current_stream()))
torch.cuda.synchronize()
s = stamps.view(-1, 16).cpu()
s = s[s[:, 12] > 0]
r0 = s[:, 11].min()
span_us = float(s[:, 15].max() - r0) / 100.0
# shader clock from the waves themselves: (exit - entry) memtime / realtime, longest-lived waves
life_t = (s[:, 14] - s[:, 12]).double(); life_r = (s[:, 15] - s[:, 11]).double()
ghz = float((life_t / (life_r * 10.0))[life_r > 500].median())
print("waves: %d   kernel span (first entry -> last exit, s_memrealtime): %.2f us   shader clock ~ %.2f GHz" % (len(s), span_us, ghz))
def q(v, scale=1.0):
    v = v.double() * scale
    return "min %7.2f  p10 %7.2f  med %7.2f  p90 %7.2f  max %7.2f us" % (v.min(), v.quantile(0.1), v.median(), v.quantile(0.9), v.max())
print("entry since first entry (realtime):", q(s[:, 11] - r0, 0.01))
print("tables staged - entry:             ", q(s[:, 13] - s[:, 12], 1e-3 / ghz))
print("exit since first entry (realtime): ", q(s[:, 15] - r0, 0.01))
print("wave lifetime:                     ", q(s[:, 14] - s[:, 12], 1e-3 / ghz))
has = s[s[:, 0] > 0]
print("waves with the stamped tile: %d" % len(has))
if len(has):
    k = 1e-3 / ghz
    print("  tile start - staged:             ", q(has[:, 0] - has[:, 13], k))
    if (has[:, 1] > 0).all():
        names = ["load+window", "pass 1", "twiddle", "exchange", "pass 2", "untangle+P", "mel", "store"]
        for i, n in enumerate(names):
            print("  %-32s " % n, q(has[:, i + 1] - has[:, i], k))
    print("  tile                             ", q(has[:, 8] - has[:, 0], k))
    print("  exit - tile end                  ", q(has[:, 14] - has[:, 8], k))
