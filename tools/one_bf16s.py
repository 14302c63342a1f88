"""one GEMM shape on the bf16-storage kernel, 6 launches (PMC aid).  usage: one_bf16s.py M K N"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lidbox_amd import _native as nv
M, K, N = (int(v) for v in sys.argv[1:4])
a16 = torch.randn(M, K, device="cuda").bfloat16(); b16 = torch.randn(N, K, device="cuda").bfloat16()
c = torch.empty(M, N, device="cuda"); c16 = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
st = nv.current_stream()
for _ in range(6):
    nv.check(nv.lib.lidbox_gemm_bf16s_nt(nv.Rows(a16.data_ptr(), 0, K, 1, M), nv.ptr(b16), K, nv.Rows(c.data_ptr(), 0, N, 1, M), nv.ptr(c16), K, N, 0, None, None, 0, st))
torch.cuda.synchronize()
