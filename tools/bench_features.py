"""Micro-benchmark of the fused log-mel kernel: achieved algorithmic GB/s vs the 8 TB/s HBM roofline."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lidbox_amd.features import audio
from lidbox_amd import _native as nv

def main():
    for B in (256, 1024, 2048):
        x = torch.randn(B, 32000, device="cuda") * 0.1
        plan = audio.get_plan(16000, 400, 160)
        for kind, name, ch in ((nv.FEAT_LOGMEL, "logmel", 40), (nv.FEAT_MFCC, "mfcc", 12), (nv.FEAT_SPECTROGRAM, "spec", 257)):
            out = plan.run(kind, x)
            torch.cuda.synchronize()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 50
            ev0.record()
            for _ in range(n):
                plan.run(kind, x, out=out)
            ev1.record()
            torch.cuda.synchronize()
            ms = ev0.elapsed_time(ev1) / n
            bytes_ = B * (32000 * 4 + 198 * ch * 4)
            print("B=%d %-6s %.3f ms  %.1f GB/s algorithmic (%.1f%% of 8 TB/s)  %.2f M utt/s" %
                  (B, name, ms, bytes_ / ms / 1e6, 100 * bytes_ / ms / 1e6 / 8000, B / ms / 1e3))

if __name__ == "__main__":
    main()
