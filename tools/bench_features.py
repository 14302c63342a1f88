"""Micro-benchmark of the fused log-mel kernel: achieved algorithmic GB/s vs the 8 TB/s HBM roofline."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lidbox_amd.features import audio
from lidbox_amd import _native as nv

def main():
    for B in (256, 512, 1024, 2048):
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

if __name__ == "__main__" and "--norm" not in sys.argv:
    main()


def norm_ops():
    """the normalisation rows of SURVEY 8a (a7-a10) on [B, 198, C] feature batches: algorithmic bytes = read + write"""
    from lidbox_amd import features
    for B, C in ((256, 40), (2048, 40), (2048, 12)):
        x = torch.randn(B, 198, C, device="cuda")
        nbytes = 2 * x.numel() * 4
        for name, fn in (("cmvn", lambda: features.cmvn(x)), ("cmn", lambda: features.cmn(x)),
                         ("window_norm(w=100)", lambda: features.window_normalization(x, window_len=100)),
                         ("feature_scaling", lambda: features.feature_scaling(x, -1.0, 1.0)),
                         ("power_to_db", lambda: audio.power_to_db(x.abs() + 1e-3))):
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 20 * 1e3
            print("B=%-5d C=%-3d %-20s %8.1f us  %7.1f GB/s algorithmic (%4.1f %% of 8 TB/s)" %
                  (B, C, name, us, nbytes / us / 1e3, nbytes / us / 1e3 / 80))


if __name__ == "__main__" and "--norm" in sys.argv:
    norm_ops()
