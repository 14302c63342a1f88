"""HBM throughput of the signal kernels (csrc/signal.hip) on a synthetic ragged batch.
usage: python tools/bench_signal.py [B] [seconds_per_utt]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lidbox_amd.features import signal_ops as sg

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
SEC = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
REPS = 20


def timeit(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REPS * 1e3


def main():
    rng = np.random.default_rng(0)
    N = int(16000 * SEC)
    x = torch.randn(B, N, device="cuda") * 0.1
    # ~30 % of the 10 ms frames are near silent
    mask = (torch.rand(B, N // 160, device="cuda") < 0.3).repeat_interleave(160, dim=1)
    x[:, :mask.shape[1]][mask] *= 1e-4
    r = sg.RaggedSignals.from_dense(x)
    nbytes = B * N * 4
    res = {}
    us = timeit(lambda: sg.frame_rms(r, 160))
    res["frame_rms (host wrapper incl. CSR upload)"] = (us, nbytes)
    vad = sg.vad_decisions(r, 160, 30, 0.1)
    kept = float(vad["decisions"].float().mean())
    us = timeit(lambda: sg.vad_decisions(r, 160, 30, 0.1))
    res["vad_decisions (rms + threshold + decide + scan)"] = (us, nbytes)
    us = timeit(lambda: sg.apply_vad(r, vad))
    res["apply_vad (incl. count read-back), kept %.0f %%" % (100 * kept)] = (us, nbytes * kept * 2)
    us = timeit(lambda: sg.signal_chunks(r, 16000, 1000, 500, 0))
    ch, _ = sg.signal_chunks(r, 16000, 1000, 500, 0)
    res["signal_chunks 1 s / 0.5 s"] = (us, nbytes + ch.numel() * 4)
    noise = torch.randn(B, N, device="cuda") * 0.01
    snr = torch.full((B,), 10.0, device="cuda")
    us = timeit(lambda: sg.snr_mixer(x, noise, snr))
    res["snr_mixer (2 reads + 3 writes)"] = (us, nbytes * 5)
    us = timeit(lambda: sg.peak_normalize(r, -3.0))
    res["peak_normalize"] = (us, nbytes * 2)
    for k, (us, by) in res.items():
        print("%-58s %9.1f us  %7.1f GB/s algorithmic (%4.1f %% of 8 TB/s)" % (k, us, by / us / 1e3, by / us / 1e3 / 80))
    print("batch: %d utterances x %.1f s = %.1f MB" % (B, SEC, nbytes / 1e6))


if __name__ == "__main__":
    main()
