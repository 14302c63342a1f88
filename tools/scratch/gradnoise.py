"""per-tensor difference of the bs-2048 gradient between GEMM kernel families (roundoff amplification through ReLU masks)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
from test_fullsize_gpu import _batch, T, MEL, SR
from lidbox_amd import _native as nv
from lidbox_amd.features import audio
from lidbox_amd.models import xvector
from lidbox_amd.train import Trainer, shard_bounds
sig, y = _batch(2048)
m = xvector.create((T, MEL), 4, seed=0)
tr = Trainer(m, feature=dict(plan=audio.get_plan(SR, 400, 160), kind=nv.FEAT_LOGMEL), use_graph=False)
def grads(env):
    for k in ("LIDBOX_GEMM_SK", "LIDBOX_GEMM_DMA", "LIDBOX_GEMM_NO_TUNED"): os.environ.pop(k, None)
    os.environ.update(env)
    _, g = tr.loss_and_grads(sig, y); g = g.double().clone()
    acc = torch.zeros_like(g)
    for r in range(8):
        lo, hi = shard_bounds(2048, r, 8)
        _, gs = tr.loss_and_grads(sig[lo:hi], y[lo:hi]); acc += gs.double()
    return g, acc / 8
fam = {"default": {}, "sk0": {"LIDBOX_GEMM_SK": "0"}, "dma0sk0": {"LIDBOX_GEMM_SK": "0", "LIDBOX_GEMM_DMA": "0"}, "notuned": {"LIDBOX_GEMM_NO_TUNED": "1"}}
G = {k: grads(v) for k, v in fam.items()}
def cmp(a, b, tag):
    worst = max(((float((a[o:o+int(np.prod(s))] - b[o:o+int(np.prod(s))]).abs().max() / (b[o:o+int(np.prod(s))].abs().max() + 1e-30)), n) for n, (o, s) in m.layout.items()))
    print("%-40s norm-rel %.3e   worst tensor max-rel %.3e (%s)" % (tag, float((a - b).norm() / b.norm()), worst[0], worst[1]))
for k in fam:
    cmp(G[k][1], G[k][0], "%s: shards vs global" % k)
for k in ("sk0", "dma0sk0", "notuned"):
    cmp(G[k][0], G["default"][0], "global: %s vs default" % k)
    cmp(G[k][1], G["default"][1], "shards: %s vs default" % k)
