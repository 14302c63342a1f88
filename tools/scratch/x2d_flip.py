"""is the xvector_2d F=64 gradient mismatch one flipped ReLU (a whole output-channel column differs) or spread out?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from oracle import torch_ref as tr
from test_xvector2d_gpu import _oracle_params, _dev
from lidbox_amd.models import xvector_2d
from lidbox_amd.train import Trainer
rng = np.random.default_rng(5)
B, T, F = 4, 21, 64
m = xvector_2d.create((T, F), 3, seed=7)
w = m.get_weights()
for k in w:
    if k.endswith("gamma"): w[k] = rng.uniform(0.5, 1.5, w[k].shape)
    if k.endswith("beta") or k.endswith(".b"): w[k] = rng.standard_normal(w[k].shape) * 0.1
m.set_weights(w)
x = rng.standard_normal((B, T, F)); y = rng.integers(0, 3, size=B).astype(np.int32)
p = _oracle_params(m)
pt = tr.to_torch_params({k: v for k, v in p.items()}, True, torch.float64)
loss_ref = tr.sparse_ce_from_logits(tr.xvector_2d_fwd(pt, torch.tensor(x), training=True), torch.tensor(y.astype(np.int64)))
loss_ref.backward()
t = Trainer(m, use_graph=False)
t.loss_and_grads(_dev(x), _dev(y, np.int32))
for k in m.layout:
    ref_g = pt[k].grad.numpy(); got = m.param(k, grad=True).cpu().numpy()
    d = np.abs(got - ref_g); rel = d.max() / max(1e-12, np.abs(ref_g).max())
    line = "%-28s max rel %.2e" % (k, rel)
    if d.ndim == 4:
        per_c = d.reshape(-1, d.shape[-1]).max(0)
        o = np.argsort(per_c)[::-1]
        line += "   worst out-channels %s : %s   median channel %.2e" % (o[:3], per_c[o[:3]], np.median(per_c))
    print(line)
