"""Fused log-mel kernel alone, for PMC collection. usage: feat_one.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lidbox_amd.features import audio
from lidbox_amd import _native as nv
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
x = torch.randn(B, 32000, device="cuda") * 0.1
plan = audio.get_plan(16000, 400, 160)
out = plan.run(nv.FEAT_LOGMEL, x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): plan.run(nv.FEAT_LOGMEL, x, out=out)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print("B=%d logmel %.3f ms  %.1f GB/s  %.2f M utt/s" % (B, ms, B * 159680 / ms / 1e6, B / ms / 1e3))
