import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from lidbox_amd import _native as nv
from lidbox_amd.models import xvector_2d
from lidbox_amd.train import Trainer
seen = set()
for name, kind in (("lidbox_gemm_nn", 0), ("lidbox_gemm_nt", 1)):
    orig = getattr(nv.lib, name)
    def w(*a, _o=orig, _k=kind, _n=name):
        A, K, N = a[0], a[4], a[5]
        M = A.batch * A.rows_per_batch
        out = (ctypes.c_int * 4)()
        nv.lib.lidbox_gemm_plan_query(_k, M, N, K, int(a[9] or 0), out)
        key = (_n, M, N, K, a[6], tuple(out), A.batch, A.rows_per_batch, A.batch_stride, A.row_stride, a[3].batch_stride, a[3].row_stride, a[2])
        if key not in seen:
            seen.add(key); print(key)
        return _o(*a)
    setattr(nv.lib, name, w)
rng = np.random.default_rng(5)
B, T, F = 4, 21, 64
m = xvector_2d.create((T, F), 3, seed=7)
t = Trainer(m, use_graph=False)
x = torch.from_numpy(rng.standard_normal((B, T, F)).astype(np.float32)).cuda()
y = torch.from_numpy(rng.integers(0, 3, size=B).astype(np.int32)).cuda()
t.loss_and_grads(x, y)
