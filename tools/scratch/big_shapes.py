"""SK / DMA kernels at the bs-2048 layer shapes vs float64 (torch fp64 on the GPU as the scratch checker)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lidbox_amd import _native as nv
st = nv.current_stream()
big = torch.empty(3 << 30, dtype=torch.uint8, device="cuda"); big.fill_(0xAB)
def R(t, M, ld): return nv.Rows(t.data_ptr(), 0, ld, 1, M)
torch.manual_seed(0)
for B in (256, 2048):
    for name, M, K, N in [("frame1", B * 198, 200, 512), ("frame2", B * 99, 1536, 512), ("frame3", B * 33, 1536, 512), ("frame4", B * 33, 512, 512), ("frame5", B * 33, 512, 1500)]:
        A = torch.randn(M, K, device="cuda"); W = torch.randn(K, N, device="cuda"); bias = torch.randn(N, device="cuda")
        # dy: correlated rows with cancellation in the column sums (as the stats-pooling backward makes them)
        DY = torch.randn(M, N, device="cuda") * 0.1 + torch.randn(1, N, device="cuda") * torch.randn(M, 1, device="cuda")
        Y = torch.zeros(M, N, device="cuda"); dW = torch.zeros(K, N, device="cuda"); db = torch.zeros(N, device="cuda")
        ref = torch.relu(A.double() @ W.double() + bias.double())
        refW = A.double().t() @ DY.double(); refb = DY.double().sum(0)
        for fam, env in (("default", {}), ("sk0", {"LIDBOX_GEMM_SK": "0"})):
            for k in ("LIDBOX_GEMM_SK",): os.environ.pop(k, None)
            os.environ.update(env)
            wsb = max(nv.lib.lidbox_gemm_rows_workspace(M, N, K), nv.lib.lidbox_gemm_tn_workspace(M, K, N))
            assert wsb <= big.numel(), wsb
            nv.check(nv.lib.lidbox_gemm_nn(R(A, M, K), nv.ptr(W), N, R(Y, M, N), K, N, nv.EPI_BIAS_RELU, nv.ptr(bias), nv.ptr(big), wsb, st))
            e1 = float((Y.double() - ref).abs().max() / ref.abs().max())
            nv.check(nv.lib.lidbox_gemm_tn(R(A, M, K), R(DY, M, N), nv.ptr(dW), N, K, N, 0, nv.ptr(db), nv.ptr(big), wsb, st))
            e2 = float((dW.double() - refW).abs().max() / refW.abs().max())
            e3 = float((db.double() - refb).abs().max() / refb.abs().max())
            print("B=%4d %s %-8s sk(nn)=%d sk(tn)=%d  fwd %.2e  dW %.2e  db %.2e" % (B, name, fam, nv.lib.lidbox_gemm_plan_is_stream_k(0, M, N, K, wsb), nv.lib.lidbox_gemm_plan_is_stream_k(2, M, N, K, wsb), e1, e2, e3))
        del A, W, DY, Y, ref, refW
