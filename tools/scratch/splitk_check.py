import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from lidbox_amd import _native as nv
st = nv.current_stream()
rng = np.random.default_rng(0)
def run(batch, rpb, bs, rs, K, N, plan, epi):
    os.environ["LIDBOX_GEMM_PLAN"] = plan
    M = batch * rpb
    X = rng.standard_normal((batch - 1) * bs + (rpb - 1) * rs + K + 64).astype(np.float32)
    W = (rng.standard_normal((K, N)) * 0.1).astype(np.float32); bias = rng.standard_normal(N).astype(np.float32)
    x, w, bi = torch.from_numpy(X).cuda(), torch.from_numpy(W).cuda(), torch.from_numpy(bias).cuda()
    y = torch.full((M, N), 3.0, device="cuda")
    ws = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    A = nv.Rows(x.data_ptr(), bs, rs, batch, rpb)
    nv.check(nv.lib.lidbox_gemm_nn(A, nv.ptr(w), N, nv.Rows(y.data_ptr(), 0, N, 1, M), K, N, epi, nv.ptr(bi), nv.ptr(ws), ws.numel(), st))
    win = np.stack([X[b * bs + t * rs: b * bs + t * rs + K] for b in range(batch) for t in range(rpb)]).astype(np.float64)
    ref = win @ W.astype(np.float64) + bias
    if epi == nv.EPI_BIAS_RELU: ref = np.maximum(ref, 0)
    err = np.abs(y.cpu().numpy() - ref).max() / np.abs(ref).max()
    print("M=%d N=%d K=%d plan %s epi %d fam %d: rel err %.2e" % (M, N, K, plan, epi, nv.lib.lidbox_gemm_last_family(), err))
for plan in ("64,64,1", "64,64,2", "64,64,4", "64,64,3", "128,64,4"):
    run(84, 29, 15360, 512, 768, 128, plan, 2)
    run(84, 9, 3712, 384, 384, 64, plan, 2)
    run(1, 2436, 0, 768, 768, 128, plan, 2)
