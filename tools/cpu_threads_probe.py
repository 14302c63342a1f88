"""How does the torch-CPU restatement scale with threads on this host? (picks cpu_baseline's thread count)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.torch_ref import TrainStepCPU
from lidbox_amd.testutil import synthetic_batch
sig, y = synthetic_batch(32, 4)
sig_t, y_t = torch.from_numpy(sig), torch.from_numpy(y.astype(np.int64))
print("cpu_count", os.cpu_count())
for th in (8, 16, 32, 64, 128):
    if th > (os.cpu_count() or 1): break
    st = TrainStepCPU(4, 0, threads=th)
    st.step(sig_t, y_t)
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < 4 and n < 20:
        st.step(sig_t, y_t); n += 1
    dt = time.perf_counter() - t0
    print("threads %d: %.2f utt/s (%d steps, %.2f s/step)" % (th, n * 32 / dt, n, dt / n), flush=True)
