"""diagnostic: per-tensor error of the captured step vs the float64 oracle at bench shape (config 1 / 3 / 4)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_fullsize_gpu as tf
from lidbox_amd import _native as nv
from lidbox_amd.features import audio
from lidbox_amd.models import cnn, xvector
from lidbox_amd.train import Trainer
config = int(sys.argv[1]) if len(sys.argv) > 1 else 3
graph = (sys.argv[2] != "eager") if len(sys.argv) > 2 else True
plan = audio.get_plan(tf.SR, 400, 160)
B, langs = 256, 4
if config == 3:
    m = cnn.create((tf.T, 12), langs, seed=0); feat = dict(plan=plan, kind=nv.FEAT_MFCC, cmvn=True)
else:
    m = xvector.create((tf.T, tf.MEL), langs, seed=0); feat = dict(plan=plan, kind=nv.FEAT_LOGMEL)
tr = Trainer(m, feature=feat, use_graph=graph)
rng = np.random.default_rng(11)
m.set_weights({k: rng.standard_normal(v.shape) * 0.05 for k, v in m.get_weights().items() if k.endswith(".b")})
w0 = {k: v.copy() for k, v in m.get_weights().items()}
sig, y = tf._batch(B, langs=langs, seed=4321)
loss = float(tr.train_step(sig, y))
got = {k: m.param(k, grad=True).cpu().numpy() for k in w0}
x_gpu = m.workspace(B, tf.T).input_view().cpu().numpy()
ref_loss, ref_g, _ = tf._oracle_step64(config, w0, x_gpu, y.cpu().numpy(), langs)
print("loss", loss, ref_loss, abs(loss - ref_loss) / abs(ref_loss))
for name, g in ref_g.items():
    d = np.abs(got[name] - g)
    i = np.unravel_index(np.argmax(d), d.shape)
    print("%-12s shape %-16s maxerr/maxabs %.2e  norm-rel %.2e  argmax %s got %.4e ref %.4e  frac>1e-3*max %.2e" % (
        name, g.shape, d.max() / np.abs(g).max(), np.linalg.norm(got[name] - g) / np.linalg.norm(g), i, got[name][i], g[i],
        (d > 1e-3 * np.abs(g).max()).mean()))
# ReLU decisions: the captured step's activations against the float64 forward pass on the same features
import torch.nn.functional as F
from oracle import torch_ref as tref, model_np
ws = m.workspace(B, tf.T)
p64 = {k: torch.tensor(np.asarray(v, dtype=np.float64)) for k, v in w0.items()}
h = torch.from_numpy(x_gpu.astype(np.float64))
layers = model_np.CNN_CONVS if config == 3 else model_np.XVECTOR_FRAMES
for i, (name, f, k, s_) in enumerate(layers):
    z = tref.conv1d_causal(h, p64[name + ".W"], p64[name + ".b"], s_, relu=False)
    h = F.relu(z)
    a_gpu = ws.act[i + 1][:, ws.pads[i + 1]:ws.pads[i + 1] + ws.Ts[i + 1], :].cpu().numpy() if i + 1 < len(ws.act) else None
    if a_gpu is None:
        break
    zz = z.numpy()
    flips = ((a_gpu > 0) != (zz > 0))
    print("%-8s outputs %9d  |z| < 1e-6: %5d  ReLU decisions that differ: %4d  (|z| there: max %.2e)  max|act - relu(z)| %.2e" % (
        name, zz.size, int((np.abs(zz) < 1e-6).sum()), int(flips.sum()), float(np.abs(zz[flips]).max()) if flips.any() else 0.0,
        float(np.abs(a_gpu - np.maximum(zz, 0)).max())))
