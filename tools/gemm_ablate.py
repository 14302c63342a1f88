"""Ablation timing of one GEMM shape under LIDBOX_GEMM_DBG / LIDBOX_GEMM_TILE (results are wrong by design)."""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, torch
from lidbox_amd import _native as nv
M,K,N = 25344,1536,512
x = torch.randn(M, K, device="cuda"); W = torch.randn(K, N, device="cuda"); y = torch.zeros(M, N, device="cuda")
A = nv.Rows(x.data_ptr(), 0, K, 1, M); Y = nv.Rows(y.data_ptr(), 0, N, 1, M); st = nv.current_stream()
f = lambda: nv.check(nv.lib.lidbox_gemm_nn(A, nv.ptr(W), N, Y, K, N, 0, None, None, 0, st))
f(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): f()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 100
print("%8.1f us  %6.1f TF/s" % (us, 2.0*M*K*N/us/1e6))
'''
for tile in ("128x128", "128x64", "64x64"):
    for dbg, what in ((0, "full"), (1, "no global loads"), (3, "no global loads, no LDS stores"), (7, "+ no barrier"),
                      (15, "no MFMA/ds_read at all"), (11, "loads/stores off, no mma, barrier only"), (8, "no mma (loads+stores+barrier)")):
        env = dict(os.environ, LIDBOX_GEMM_DBG=str(dbg), LIDBOX_GEMM_TILE=tile)
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True,
                             cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        print("%-8s dbg=%-2d %-40s %s" % (tile, dbg, what, out.stdout.strip() or out.stderr.strip()[-200:]), flush=True)
