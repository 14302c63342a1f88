"""Throughput of the BASELINE.json configurations that bench.py does not print (bench.py's line is configs[1] / [2]):

  configs[3]  MFCC(1:13) + CMVN front-end -> lidbox.models.cnn spectrogram classifier, bs 256, fp32, 1 GPU
  configs[4]  100-language x-vector trunk -> segment1 -> L2 norm -> SparseAngularProximity + C_avg (100 thresholds),
              bf16 compute / fp32 master, ONE GPU's shard of the 8 x 512 batch (bs 512)

  x2d         (not a BASELINE config) log-mel -> xvector_2d (Conv2D-along-frequency + BatchNormalization front-end
              -> x-vector), bs 256, fp32: forward 1 043 MFLOP / utterance (front-end 731), train ~ 3 090

Same timing discipline as bench.py: inputs resident in HBM, warm-up, K graph-replayed steps between synchronisations.
Algorithmic train flops per utterance: CNN 2 135 MFLOP, x-vector (100 outputs replaced by the 512-d AP head) 918 MFLOP
(SURVEY 8d).  usage: python tools/bench_configs.py [--steps K] [--warmup W]"""
import argparse
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lidbox_amd import _native as nv
from lidbox_amd.features import audio
from lidbox_amd.losses import SparseAngularProximity
from lidbox_amd.metrics import SparseAverageDetectionCost
from lidbox_amd.models import cnn, xvector, xvector_2d
from lidbox_amd.models.tdnn import DenseSpec, SequentialTDNN
from lidbox_amd.testutil import synthetic_batch
from lidbox_amd.train import Trainer


def run(name, trainer, sig, lab, steps, warmup, flops_per_utt, extra):
    for _ in range(warmup):
        trainer.train_step(sig, lab)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = trainer.train_step(sig, lab)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    B = sig.shape[0]
    out = dict(config=name, value=round(B * steps / dt, 1), unit="utterances/s", ms_per_step=round(1e3 * dt / steps, 4),
               batch=B, steps=steps, warmup=warmup, step_tflops=round(B * steps / dt * flops_per_utt / 1e12, 2),
               final_loss=round(float(loss), 6))
    out.update({k: (v() if callable(v) else v) for k, v in extra.items()})
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--only", choices=["cnn", "ap", "ap32", "x2d"], default=None)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    plan = audio.get_plan(16000, 400, 160, device=dev)
    if a.only in (None, "cnn"):
        sig, y = synthetic_batch(256, 4, 16000, 2.0, seed=1234)
        m = cnn.create((198, 12), 4, seed=0, device=dev)
        t = Trainer(m, feature=dict(plan=plan, kind=nv.FEAT_MFCC, cmvn=True), use_graph=True)
        run("configs[3]: MFCC(1:13)+CMVN -> cnn, 4 languages, bs 256, fp32, 1 GPU", t, torch.from_numpy(sig).to(dev),
            torch.from_numpy(y.astype(np.int32)).to(dev), a.steps, a.warmup, 2135e6, dict(dtype="f32"))
        del t, m
    if a.only in (None, "x2d"):
        sig, y = synthetic_batch(256, 4, 16000, 2.0, seed=1234)
        m = xvector_2d.create((198, 40), 4, seed=0, device=dev)
        t = Trainer(m, feature=dict(plan=plan, kind=nv.FEAT_LOGMEL), use_graph=True)
        run("8f.1: log-mel -> xvector_2d (2-D front-end + BatchNorm), 4 languages, bs 256, fp32, 1 GPU", t, torch.from_numpy(sig).to(dev),
            torch.from_numpy(y.astype(np.int32)).to(dev), a.steps, a.warmup, 3090e6, dict(dtype="f32"))
        del t, m
        torch.cuda.empty_cache()
    for tag, cd in (("ap", "bfloat16"), ("ap32", "float32")):
        if a.only not in (None, tag):
            continue
        N, D, B = 100, 512, 512
        sig, y = synthetic_batch(B, N, 16000, 2.0, seed=1234)
        convs = [xvector.frame_layer(512, 5, 1, name="frame1"), xvector.frame_layer(512, 3, 2, name="frame2"),
                 xvector.frame_layer(512, 3, 3, name="frame3"), xvector.frame_layer(512, 1, 1, name="frame4"),
                 xvector.frame_layer(1500, 1, 1, name="frame5")]
        m = SequentialTDNN((198, 40), convs, "stats", [DenseSpec("segment1", D, relu=False)], output_activation=None, seed=0,
                           device=dev, compute_dtype=cd)
        metric = SparseAverageDetectionCost(N, np.linspace(-np.pi, 0, 100))
        t = Trainer(m, loss=SparseAngularProximity(N, D), feature=dict(plan=plan, kind=nv.FEAT_LOGMEL), use_graph=True, metric=metric)
        run("configs[4] shard: log-mel -> x-vector trunk -> AP loss + C_avg, 100 languages, bs 512 per GPU, %s" % cd,
            t, torch.from_numpy(sig).to(dev), torch.from_numpy(y.astype(np.int32)).to(dev), a.steps, a.warmup, 915.6e6,
            dict(dtype="bf16" if cd == "bfloat16" else "f32", c_avg=lambda metric=metric: round(float(metric.result()), 4)))
        del t, m


if __name__ == "__main__":
    main()
