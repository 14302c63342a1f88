"""Debug: per-phase s_memtime breakdown of the fused feature kernel.  Needs a library built with
`python tools/ab_build.py feattiming features.hip -DLBX_FEAT_TIMING`; run as
`LIDBOX_HIP_LIB=tools/ab/libfeattiming.so python tools/feat_phases.py [B]`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lidbox_amd.features import audio
from lidbox_amd import _native as nv
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
x = torch.randn(B, 32000, device="cuda") * 0.1
plan = audio.get_plan(16000, 400, 160)
out = torch.empty(B, 198, 40, device="cuda")
stamps = torch.zeros(4096 * 4 * 12, dtype=torch.int64, device="cuda")
for _ in range(3):
    nv.check(nv.lib.lidbox_extract_features_fwd(plan.handle, nv.FEAT_LOGMEL, nv.ptr(x), B, 32000, 32000, nv.ptr(out), 0,
                                                nv.ptr(stamps), stamps.numel() * 8, nv.current_stream()))
torch.cuda.synchronize()
s = stamps.view(-1, 12).cpu()
s = s[s[:, 0] > 0][:, :9].double()
d = (s[:, 1:] - s[:, :-1])
names = ["load+window", "dft16 x2 (pass 1)", "twiddle", "exchange", "dft16 x2 (pass 2)", "untangle+P", "mel", "store/mfcc"]
print("waves sampled:", len(s), " total per tile (median): %.0f ticks" % (s[:, 8] - s[:, 0]).median())
for i, n in enumerate(names):
    print("  %-20s median %8.0f  mean %8.0f" % (n, d[:, i].median(), d[:, i].mean()))
