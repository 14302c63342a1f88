"""
oracle.torch_ref -- torch-CPU fp32 restatement of the same math (autograd).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Two uses:
  * an independent backward (torch autograd) to cross-check model_np's
    hand-derived gradients and the HIP backward kernels;
  * the timed `cpu_baseline` ("port") in bench.py: log-mel + x-vector train
    step on the host cores, multi-threaded (the stand-in for "lidbox's own TF
    CPU path", which cannot run here -- BASELINE.md section 4).

All file:line citations are relative to /root/reference/.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import features_np, model_np


# ---------------------------------------------------------------- features
class LogMelCPU:
    """lidbox/data/tf_utils.py:166-179 for feattype="logmelspectrogram" with defaults."""

    def __init__(self, sample_rate=16000, frame_length_ms=25, frame_step_ms=10, fft_length=512,
                 num_mel_bins=40, fmin=0.0, fmax=8000.0):
        self.L = features_np.ms_to_frames(sample_rate, frame_length_ms)
        self.S = features_np.ms_to_frames(sample_rate, frame_step_ms)
        self.nfft = fft_length
        self.window = torch.from_numpy(features_np.hann_window(self.L, True, np.float32))
        self.W = torch.from_numpy(features_np.linear_to_mel_weight_matrix(
            num_mel_bins, fft_length // 2 + 1, sample_rate, fmin, fmax, np.float32))

    def __call__(self, signals):
        fr = signals.unfold(-1, self.L, self.S) * self.window          # tf.signal.frame + window
        S = torch.fft.rfft(fr, n=self.nfft, dim=-1)                    # right zero-pad to fft_length
        P = S.real * S.real + S.imag * S.imag                          # |.|^2   audio.py:230
        return torch.log(P @ self.W + 1e-6)                            # audio.py:261, tf_utils.py:178


# ---------------------------------------------------------------- model
def conv1d_causal(x, W, b, s, relu=True, padding="causal"):
    """x [B,T,C]; W Keras layout [k,C_in,C_out]; zero rows per model_np.conv1d_padding."""
    k = W.shape[0]
    pl, pr, To = model_np.conv1d_padding(x.shape[1], k, s, 1, padding)
    if To == 0:
        return x.new_zeros((x.shape[0], 0, W.shape[2]))
    y = F.conv1d(F.pad(x.transpose(1, 2), (pl, pr)), W.permute(2, 1, 0), b, stride=s)
    y = y.transpose(1, 2)
    return F.relu(y) if relu else y


def stats_pool(x):
    """lidbox/models/xvector.py:30-35."""
    mean = x.mean(dim=1, keepdim=True)
    var = ((x - mean) ** 2).mean(dim=1)
    std = torch.sqrt(torch.clamp(var, min=model_np.STDDEV_SQRT_MIN_CLIP))
    return torch.cat([mean[:, 0, :], std], dim=1)


def xvector_fwd(p, x, embedding=False):
    h = x
    for name, f, k, s in model_np.XVECTOR_FRAMES:
        h = conv1d_causal(h, p[name + ".W"], p[name + ".b"], s)
    h = stats_pool(h)
    if embedding:
        return h @ p["segment1.W"] + p["segment1.b"]
    h = F.relu(h @ p["segment1.W"] + p["segment1.b"])
    h = F.relu(h @ p["segment2.W"] + p["segment2.b"])
    z = h @ p["outputs.W"] + p["outputs.b"]
    return F.log_softmax(z, dim=-1)


def cnn_fwd(p, x, padding="causal"):
    h = x
    for name, f, k, s in model_np.CNN_CONVS:
        h = conv1d_causal(h, p[name + ".W"], p[name + ".b"], s, padding=padding)
    h = h.mean(dim=1)
    h = F.relu(h @ p["fc_1.W"] + p["fc_1.b"])
    h = F.relu(h @ p["fc_2.W"] + p["fc_2.b"])
    return F.log_softmax(h @ p["output.W"] + p["output.b"], dim=-1)


def sparse_ce_from_logits(outputs, y):
    """Keras SparseCategoricalCrossentropy(from_logits=True), mean reduction."""
    return F.cross_entropy(outputs, y)


def ap_loss(y, z, N, delta_weight=1.0):
    """lidbox/losses.py:25-49, mean reduction."""
    th = torch.acos(z[:, :N])
    th_l = th.gather(1, y[:, None])
    sig = torch.sigmoid(delta_weight * (th_l - th))
    mask = 1.0 - F.one_hot(y, N).to(z.dtype)
    return (mask * sig).sum(dim=1).mean()


def to_torch_params(p_np, requires_grad=True, dtype=torch.float32):
    return {k: torch.tensor(v, dtype=dtype, requires_grad=requires_grad) for k, v in p_np.items()}


class KerasAdam:
    """tf.keras.optimizers.Adam defaults (lr 1e-3, betas 0.9/0.999, epsilon 1e-7)."""

    def __init__(self, params, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-7):
        self.params, self.lr, self.b1, self.b2, self.eps = params, lr, beta1, beta2, eps
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}
        self.t = 0

    @torch.no_grad()
    def step(self):
        self.t += 1
        lr_t = self.lr * math.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t)
        for k, p in self.params.items():
            g = p.grad
            self.m[k].mul_(self.b1).add_(g, alpha=1.0 - self.b1)
            self.v[k].mul_(self.b2).addcmul_(g, g, value=1.0 - self.b2)
            p.addcdiv_(self.m[k], self.v[k].sqrt().add_(self.eps), value=-lr_t)
            p.grad = None


class TrainStepCPU:
    """train step on the host (the timed cpu_baseline of bench.py).
    config "xvector": log-mel -> x-vector -> sparse CE (BASELINE configs[1]);
           "cnn":     MFCC(1:13) + CMVN -> cnn -> sparse CE (configs[3]; tf_utils.py:180-185, features/__init__.py:22-32);
           "ap":      log-mel -> x-vector trunk -> segment1 (no activation) -> L2 normalise -> angular proximity loss
                      (configs[4]'s per-GPU shard; losses.py:25-49), fp32 on the CPU."""

    def __init__(self, num_outputs=4, seed=0, threads=None, config="xvector"):
        if threads:
            torch.set_num_threads(threads)
        self.config = config
        self.feat = LogMelCPU()
        self.num_outputs = num_outputs
        if config == "cnn":
            self.dct = torch.from_numpy(features_np.dct_matrix(40, np.float32)[:, 1:13].copy())
            self.params = to_torch_params(model_np.cnn_init(12, num_outputs, seed))
        elif config == "ap":
            p = model_np.xvector_init(40, num_outputs, seed)
            self.params = to_torch_params({k: v for k, v in p.items() if not k.startswith(("segment2", "outputs"))})
        else:
            self.params = to_torch_params(model_np.xvector_init(40, num_outputs, seed))
        self.opt = KerasAdam(self.params)

    def step(self, signals, labels):
        with torch.no_grad():
            x = self.feat(signals)
            if self.config == "cnn":
                x = x @ self.dct
                mean = x.mean(dim=1, keepdim=True)
                std = x.std(dim=1, unbiased=False, keepdim=True)
                x = torch.where(std > 0, (x - mean) / std, torch.zeros_like(x))
        if self.config == "cnn":
            loss = sparse_ce_from_logits(cnn_fwd(self.params, x), labels)
        elif self.config == "ap":
            z = F.normalize(xvector_fwd(self.params, x, embedding=True), dim=1)
            loss = ap_loss(labels, z, self.num_outputs)
        else:
            loss = sparse_ce_from_logits(xvector_fwd(self.params, x), labels)
        loss.backward()
        self.opt.step()
        return float(loss.detach())


def xvector_2d_fwd(p, x, training=True):
    """lidbox/models/xvector_2d.py:66-93 (autograd cross-check of the BatchNorm / Conv2D backward).  p: torch parameter
    dict in Keras layouts (model_np.xvector_2d_init), x [B,T,F].  Batch statistics when training."""
    B, T, Fq = x.shape
    h = x.reshape(B * T, 1, Fq)                                                   # frames are independent: [N, C_in, F]
    for name, f, k, s in model_np.FRAMES_2D:
        W = p[name + "_conv.W"][0].permute(2, 1, 0)                               # [k, C_in, C_out] -> [C_out, C_in, k]
        h = F.relu(F.conv1d(h, W, p[name + "_conv.b"], stride=s))
        if training:
            mean = h.mean(dim=(0, 2), keepdim=True)
            var = ((h - mean) ** 2).mean(dim=(0, 2), keepdim=True)
        else:
            mean = p[name + "_bn.moving_mean"][None, :, None]
            var = p[name + "_bn.moving_variance"][None, :, None]
        h = p[name + "_bn.gamma"][None, :, None] * (h - mean) / torch.sqrt(var + model_np.BN_EPSILON) \
            + p[name + "_bn.beta"][None, :, None]
    # [B*T, C, cols] -> [B, T, cols * C] (Keras flattens (cols, channels) with channels fastest)
    h = h.permute(0, 2, 1).reshape(B, T, -1)
    for name, f, k, s in model_np.XVECTOR_FRAMES:
        h = conv1d_causal(h, p[name + ".W"], p[name + ".b"], s)
    h = stats_pool(h)
    h = F.relu(h @ p["segment1.W"] + p["segment1.b"])
    h = F.relu(h @ p["segment2.W"] + p["segment2.b"])
    return F.log_softmax(h @ p["output.W"] + p["output.b"], dim=-1)
