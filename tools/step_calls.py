"""Every GEMM-family C-ABI call of one eager train step with its shape and HIP-event time (kernel + its reduce).
usage: python tools/step_calls.py [float32|bfloat16] [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lidbox_amd import _native as nv
from lidbox_amd.features import audio
from lidbox_amd.models import xvector
from lidbox_amd.testutil import synthetic_batch
from lidbox_amd.train import Trainer
dt = sys.argv[1] if len(sys.argv) > 1 else "float32"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
sig, y = synthetic_batch(B, num_labels=4, duration_s=2.0)
sd, yd = torch.from_numpy(sig).cuda(), torch.from_numpy(y.astype(np.int32)).cuda()
m = xvector.create((198, 40), 4, seed=0, compute_dtype=dt)
t = Trainer(m, feature=dict(plan=audio.get_plan(16000, 400, 160), kind=nv.FEAT_LOGMEL), use_graph=False)
for _ in range(3): t.train_step(sd, yd)
names = [n for n in dir(nv.lib) if n.startswith("lidbox_gemm") and ("_nn" in n or "_nt" in n or "_tn" in n) and "workspace" not in n]
recs = []
def wrap(name, orig):
    def f(*a):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); rc = orig(*a); e1.record()
        A = a[0]
        M = A.batch * A.rows_per_batch
        if "bf16s_nt_pair" in name: K, N = a[5] + a[14], a[6]      # two problems in one grid: their contractions side by side
        elif "bf16s_nt" in name: K, N = a[5], a[6]
        else: K, N = a[4], a[5]
        recs.append((name, M, K, N, e0, e1))
        return rc
    return f
for n in names: setattr(nv.lib, n, wrap(n, getattr(nv.lib, n)))
REP = 5
for _ in range(REP): t.train_step(sd, yd)
torch.cuda.synchronize()
per = len(recs) // REP
tot = 0.0
for i in range(per):
    us = sorted(recs[i + r * per][4].elapsed_time(recs[i + r * per][5]) * 1e3 for r in range(REP))[REP // 2]
    n, M, K, N = recs[i][:4]
    tot += us
    print("%-22s M=%6d K=%5d N=%5d  %7.1f us  %6.1f TF" % (n.replace("lidbox_gemm_", ""), M, K, N, us, 2.0 * M * K * N / us * 1e-6))
print("total %.1f us over %d calls" % (tot, per))
