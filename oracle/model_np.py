"""
oracle.model_np -- numpy restatement of the lidbox x-vector / CNN models, the
angular-proximity loss, the C_avg metric and the Keras train step.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED for values
(no TensorFlow here); Keras layer semantics are restated from their documented
behaviour next to the reference line that instantiates them.  The hand-derived
backward passes are cross-checked against torch autograd (oracle/torch_ref.py)
in tests/test_oracle.py.

All file:line citations are relative to /root/reference/.
"""
import numpy as np

STDDEV_SQRT_MIN_CLIP = 1e-10        # lidbox/models/xvector.py:22


# ------------------------------------------------------------------ initialisers
def glorot_uniform(rng, shape, fan_in, fan_out, dtype=np.float32):
    """Keras default kernel_initializer for Conv1D/Dense: U(-l, l), l = sqrt(6/(fan_in+fan_out))."""
    limit = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-limit, limit, size=shape).astype(dtype)


# ------------------------------------------------------------------ a12 Conv1D causal
def conv1d_out_len(T, stride):
    """Keras Conv1D(padding="causal"): left-pad (k-1) then VALID with stride s."""
    return (T - 1) // stride + 1 if T > 0 else 0


def conv1d_padding(T, k, s, d=1, padding="causal"):
    """(zero rows ahead, zero rows behind, output length) of Keras Conv1D(padding=...), which lidbox/models/cnn.py:25,33-36
    passes through from its `padding` argument.  TensorFlow's rules with k_eff = (k-1)*d + 1: "causal" pads k_eff - 1 rows
    ahead and runs VALID; "valid": ceil((T - k_eff + 1) / s) outputs, no padding; "same": ceil(T / s) outputs,
    pad_total = max((out - 1)*s + k_eff - T, 0), pad_before = pad_total // 2 (the odd row goes behind)."""
    ke = (k - 1) * d + 1
    if T <= 0:
        return (ke - 1 if padding == "causal" else 0), 0, 0
    if padding == "causal":
        return ke - 1, 0, conv1d_out_len(T, s)
    if padding == "valid":
        return 0, 0, max(0, -(-(T - ke + 1) // s))
    assert padding == "same", padding
    out = -(-T // s)
    total = max((out - 1) * s + ke - T, 0)
    return total // 2, total - total // 2, out


def im2col_causal(x, k, s, d=1, padding="causal"):
    """x [B,T,C] -> col [B,T_out,k*C] with col[b,t,j*C+c] = xpad[b, t*s+j*d, c],
    xpad = (k-1)*d zero rows then x.  (Keras causal padding, lidbox/models/xvector.py:38-39;
    d = Conv1D dilation_rate -- the reference never sets it (SURVEY 8f.1 "opt-in dilation"), Keras
    requires s == 1 when d > 1.)  Other paddings (cnn.py:25): xpad per conv1d_padding."""
    B, T, C = x.shape
    assert d == 1 or s == 1, "Keras Conv1D: strides > 1 not supported together with dilation_rate > 1"
    pl, pr, To = conv1d_padding(T, k, s, d, padding)
    xp = np.concatenate([np.zeros((B, pl, C), x.dtype), x, np.zeros((B, pr, C), x.dtype)], axis=1)
    idx = np.arange(To)[:, None] * s + np.arange(k)[None, :] * d
    return xp[:, idx, :].reshape(B, To, k * C)


def conv1d_causal_fwd(x, W, b, s, relu=True, d=1, padding="causal"):
    """W [k,C_in,C_out] (Keras kernel layout), b [C_out]."""
    k, Ci, Co = W.shape
    col = im2col_causal(x, k, s, d, padding)
    y = col @ W.reshape(k * Ci, Co) + b
    return np.maximum(y, 0) if relu else y


def conv1d_causal_bwd(x, W, y, dy, s, relu=True, need_dx=True, d=1, padding="causal"):
    """Returns (dx, dW, db).  y is the post-activation output."""
    k, Ci, Co = W.shape
    B, T, _ = x.shape
    if relu:
        dy = dy * (y > 0)
    col = im2col_causal(x, k, s, d, padding)
    To = col.shape[1]
    dW = (col.reshape(-1, k * Ci).T @ dy.reshape(-1, Co)).reshape(k, Ci, Co)
    db = dy.reshape(-1, Co).sum(axis=0)
    dx = None
    if need_dx:
        pl, pr, _ = conv1d_padding(T, k, s, d, padding)
        dcol = (dy @ W.reshape(k * Ci, Co).T).reshape(B, To, k, Ci)
        dxp = np.zeros((B, pl + T + pr, Ci), x.dtype)
        for j in range(k):
            dxp[:, np.arange(To) * s + j * d, :] += dcol[:, :, j, :]
        dx = dxp[:, pl:pl + T, :]
    return dx, dW, db


# ------------------------------------------------------------------ a13 stats pooling
def stats_pool_fwd(x):
    """lidbox/models/xvector.py:25-35: mean, sqrt(clip(mean((x-mean)^2), 1e-10, fmax)), concat."""
    mean = x.mean(axis=1, keepdims=True)
    var = ((x - mean) ** 2).mean(axis=1)
    std = np.sqrt(np.clip(var, STDDEV_SQRT_MIN_CLIP, np.finfo(x.dtype).max))
    return np.concatenate([mean[:, 0, :], std], axis=1)


def stats_pool_bwd(x, dout):
    """d/dx of stats_pool_fwd; tf.clip_by_value passes gradient only inside [lo, hi]."""
    B, T, C = x.shape
    mean = x.mean(axis=1, keepdims=True)
    d = x - mean
    var = (d ** 2).mean(axis=1)
    std = np.sqrt(np.clip(var, STDDEV_SQRT_MIN_CLIP, np.finfo(x.dtype).max))
    dmean, dstd = dout[:, :C], dout[:, C:]
    inside = (var >= STDDEV_SQRT_MIN_CLIP)
    dvar = np.where(inside, dstd / (2.0 * std), 0.0)
    return (dmean[:, None, :] / T + dvar[:, None, :] * 2.0 * d / T).astype(x.dtype)


def global_avg_pool_fwd(x):
    """Keras GlobalAveragePooling1D (lidbox/models/cnn.py:37)."""
    return x.mean(axis=1)


# ------------------------------------------------------------------ a14 dense, log-softmax
def dense_fwd(x, W, b, relu=True):
    y = x @ W + b
    return np.maximum(y, 0) if relu else y


def dense_bwd(x, W, y, dy, relu=True):
    if relu:
        dy = dy * (y > 0)
    return dy @ W.T, x.T @ dy, dy.sum(axis=0)


def log_softmax(z):
    """tf.nn.log_softmax over the last axis (lidbox/models/xvector.py:65)."""
    m = z.max(axis=-1, keepdims=True)
    e = z - m
    return e - np.log(np.exp(e).sum(axis=-1, keepdims=True))


def sparse_ce_from_logits(outputs, y):
    """Keras SparseCategoricalCrossentropy(from_logits=True) on the model outputs
    (which are already log-probabilities): mean_b( -log_softmax(outputs)[b, y_b] ).
    Selected by config at lidbox/models/keras_utils.py:141-142; reduction = mean."""
    lp = log_softmax(outputs)
    return float(-lp[np.arange(len(y)), y].mean())


def sparse_ce_from_logits_grad(outputs, y):
    """d loss / d outputs = (softmax(outputs) - onehot) / B."""
    p = np.exp(log_softmax(outputs))
    p[np.arange(len(y)), y] -= 1.0
    return p / len(y)


# ------------------------------------------------------------------ a15 x-vector
XVECTOR_FRAMES = [  # (name, filters, kernel, stride)   lidbox/models/xvector.py:53-57
    ("frame1", 512, 5, 1), ("frame2", 512, 3, 2), ("frame3", 512, 3, 3),
    ("frame4", 512, 1, 1), ("frame5", 1500, 1, 1)]
XVECTOR_SEGMENTS = [("segment1", 512), ("segment2", 512)]     # xvector.py:61-62


def xvector_init(input_dim, num_outputs, seed=0, dtype=np.float32):
    """Parameter dict in Keras layouts: Conv1D kernel [k,C_in,C_out], Dense [in,out]."""
    rng = np.random.default_rng(seed)
    p, c = {}, input_dim
    for name, f, k, s in XVECTOR_FRAMES:
        p[name + ".W"] = glorot_uniform(rng, (k, c, f), k * c, k * f, dtype)
        p[name + ".b"] = np.zeros(f, dtype)
        c = f
    c = 2 * c
    for name, u in XVECTOR_SEGMENTS + [("outputs", num_outputs)]:
        p[name + ".W"] = glorot_uniform(rng, (c, u), c, u, dtype)
        p[name + ".b"] = np.zeros(u, dtype)
        c = u
    return p


def xvector_fwd(p, x, return_cache=False, embedding=False):
    """lidbox/models/xvector.py:46-67 (channel_dropout_rate=0).  x [B,T,C] -> log-probs [B,N].
    embedding=True: as_embedding_extractor (:70-73) -- segment1 affine output, no ReLU."""
    acts = [x]
    h = x
    for name, f, k, s in XVECTOR_FRAMES:
        h = conv1d_causal_fwd(h, p[name + ".W"], p[name + ".b"], s, relu=True)
        acts.append(h)
    pooled = stats_pool_fwd(h)
    if embedding:
        return dense_fwd(pooled, p["segment1.W"], p["segment1.b"], relu=False)
    s1 = dense_fwd(pooled, p["segment1.W"], p["segment1.b"])
    s2 = dense_fwd(s1, p["segment2.W"], p["segment2.b"])
    z = dense_fwd(s2, p["outputs.W"], p["outputs.b"], relu=False)
    out = log_softmax(z)
    if return_cache:
        return out, dict(acts=acts, pooled=pooled, s1=s1, s2=s2, z=z)
    return out


def xvector_loss_and_grads(p, x, y):
    """One Keras train_step's forward/backward: loss = sparse CE(from_logits) on the
    log-softmax outputs, mean over the batch.  Returns (loss, grads, logp)."""
    out, c = xvector_fwd(p, x, return_cache=True)
    loss = sparse_ce_from_logits(out, y)
    dout = sparse_ce_from_logits_grad(out, y)
    # through log_softmax: dz = dout - softmax(z) * sum(dout)
    sm = np.exp(out)
    dz = dout - sm * dout.sum(axis=-1, keepdims=True)
    g = {}
    dh, g["outputs.W"], g["outputs.b"] = dense_bwd(c["s2"], p["outputs.W"], c["z"], dz, relu=False)
    dh, g["segment2.W"], g["segment2.b"] = dense_bwd(c["s1"], p["segment2.W"], c["s2"], dh)
    dh, g["segment1.W"], g["segment1.b"] = dense_bwd(c["pooled"], p["segment1.W"], c["s1"], dh)
    dh = stats_pool_bwd(c["acts"][-1], dh)
    for i in range(len(XVECTOR_FRAMES) - 1, -1, -1):
        name, f, k, s = XVECTOR_FRAMES[i]
        dh, g[name + ".W"], g[name + ".b"] = conv1d_causal_bwd(
            c["acts"][i], p[name + ".W"], c["acts"][i + 1], dh, s, relu=True, need_dx=(i > 0))
    return loss, g, out


# ------------------------------------------------------------------ a16 CNN
CNN_CONVS = [("conv_1", 500, 5, 1), ("conv_2", 500, 7, 2), ("conv_3", 500, 1, 1),
             ("conv_4", 3000, 1, 1)]                      # lidbox/models/cnn.py:32-35
CNN_DENSE = [("fc_1", 1500), ("fc_2", 600)]               # cnn.py:39-40


def cnn_init(input_dim, num_outputs, seed=0, dtype=np.float32):
    rng = np.random.default_rng(seed)
    p, c = {}, input_dim
    for name, f, k, s in CNN_CONVS:
        p[name + ".W"] = glorot_uniform(rng, (k, c, f), k * c, k * f, dtype)
        p[name + ".b"] = np.zeros(f, dtype)
        c = f
    for name, u in CNN_DENSE + [("output", num_outputs)]:
        p[name + ".W"] = glorot_uniform(rng, (c, u), c, u, dtype)
        p[name + ".b"] = np.zeros(u, dtype)
        c = u
    return p


def cnn_fwd(p, x, embedding=False, padding="causal"):
    """lidbox/models/cnn.py:25-45 with output_activation="log_softmax"."""
    h = x
    for name, f, k, s in CNN_CONVS:
        h = conv1d_causal_fwd(h, p[name + ".W"], p[name + ".b"], s, padding=padding)
    h = global_avg_pool_fwd(h)
    if embedding:                                          # cnn.py:19-22
        return dense_fwd(h, p["fc_1.W"], p["fc_1.b"], relu=False)
    h = dense_fwd(h, p["fc_1.W"], p["fc_1.b"])
    h = dense_fwd(h, p["fc_2.W"], p["fc_2.b"])
    return log_softmax(dense_fwd(h, p["output.W"], p["output.b"], relu=False))


# ------------------------------------------------------------------ 8f.1 frequency attention
def softmax(z):
    e = np.exp(z - z.max(axis=-1, keepdims=True))
    return e / e.sum(axis=-1, keepdims=True)


def freq_attention_fwd(H, Wf1, Wf2, return_cache=False):
    """lidbox/models/clstm.py:31-42 as used by xvector_freq_attention.py:29: H [B,T,d_h];
    x1 = relu(H Wf_1) (Dense d_a, no bias, :35); F_A = softmax(x1 Wf_2) over d_f bins (:36);
    the d_h channels are partitioned into d_f consecutive bins of d_h/d_f channels (:39) and bin f is
    scaled by F_A[..., f] (:38-40)."""
    B, T, C = H.shape
    d_f = Wf2.shape[1]
    assert C % d_f == 0                                                              # clstm.py:32
    x1 = np.maximum(H @ Wf1, 0)
    F = softmax(x1 @ Wf2)
    Hw = (H.reshape(B, T, d_f, C // d_f) * F[..., None]).reshape(B, T, C)
    if return_cache:
        return Hw, dict(x1=x1, F=F)
    return Hw


def freq_attention_bwd(H, Wf1, Wf2, cache, dHw):
    """Returns (dH, dWf1, dWf2) of freq_attention_fwd (dH excludes any activation mask on H)."""
    B, T, C = H.shape
    d_f = Wf2.shape[1]
    x1, F = cache["x1"], cache["F"]
    dHw_b = dHw.reshape(B, T, d_f, C // d_f)
    dF = (dHw_b * H.reshape(B, T, d_f, C // d_f)).sum(axis=-1)
    dH = (dHw_b * F[..., None]).reshape(B, T, C)
    dlogit = F * (dF - (F * dF).sum(axis=-1, keepdims=True))
    dWf2 = x1.reshape(-1, x1.shape[-1]).T @ dlogit.reshape(-1, d_f)
    dx1 = (dlogit @ Wf2.T) * (x1 > 0)
    dWf1 = H.reshape(-1, C).T @ dx1.reshape(-1, x1.shape[-1])
    dH = dH + dx1 @ Wf1.T
    return dH, dWf1, dWf2


# ------------------------------------------------------------------ a19 Adam (Keras)
def adam_step(params, grads, m, v, step, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-7):
    """tf.keras.optimizers.Adam (non-amsgrad) dense update, epsilon=1e-7:
    lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m,v EMA; theta -= lr_t*m/(sqrt(v)+eps).  step is 1-based."""
    lr_t = lr * np.sqrt(1.0 - beta2 ** step) / (1.0 - beta1 ** step)
    for k in params:
        g = grads[k]
        m[k] = beta1 * m[k] + (1.0 - beta1) * g
        v[k] = beta2 * v[k] + (1.0 - beta2) * g * g
        params[k] = (params[k] - lr_t * m[k] / (np.sqrt(v[k]) + eps)).astype(params[k].dtype)


def sgd_step(params, grads, vel, lr=0.01, momentum=0.0, nesterov=False):
    """tf.keras.optimizers.SGD dense update (the class a config may name at lidbox/models/keras_utils.py:137-140; TensorFlow
    2.3 gradient_descent.py): momentum == 0: w -= lr g; else v = momentum v - lr g, w += (momentum v - lr g if nesterov else v)."""
    for k in params:
        g = grads[k]
        if momentum == 0.0:
            params[k] = (params[k] - lr * g).astype(params[k].dtype)
        else:
            vel[k] = momentum * vel[k] - lr * g
            params[k] = (params[k] + (momentum * vel[k] - lr * g if nesterov else vel[k])).astype(params[k].dtype)


def rmsprop_step(params, grads, rms, mg, mom, lr=1e-3, rho=0.9, momentum=0.0, eps=1e-7, centered=False):
    """tf.keras.optimizers.RMSprop dense update (keras_utils.py:137-140; TensorFlow 2.3 rmsprop.py:_resource_apply_dense):
    rms = rho rms + (1 - rho) g^2; centered: mg = rho mg + (1 - rho) g and denom = rms - mg^2.  Without momentum the Python
    branch: w -= lr g / (sqrt(denom) + eps); with momentum the fused training op (training_ops.apply_rms_prop):
    mom = momentum mom + lr g / sqrt(denom + eps), w -= mom -- epsilon INSIDE the root there."""
    for k in params:
        g = grads[k]
        rms[k] = rho * rms[k] + (1.0 - rho) * g * g
        denom = rms[k]
        if centered:
            mg[k] = rho * mg[k] + (1.0 - rho) * g
            denom = rms[k] - mg[k] * mg[k]
        if momentum > 0.0:
            mom[k] = momentum * mom[k] + lr * g / np.sqrt(denom + eps)
            params[k] = (params[k] - mom[k]).astype(params[k].dtype)
        else:
            params[k] = (params[k] - lr * g / (np.sqrt(denom) + eps)).astype(params[k].dtype)


def sparse_ce_from_probs(z, y, eps=1e-7):
    """Keras SparseCategoricalCrossentropy(from_logits=False) on the outputs of a model that ends in tf.nn.softmax
    (lidbox/models/cnn.py:43-44 with output_activation="softmax", loss class from keras_utils.py:141-142).  TensorFlow 2.3
    backend.sparse_categorical_crossentropy: output = clip_by_value(output, eps, 1 - eps); sparse_softmax_cross_entropy_with_
    logits(labels, log(output)) -- i.e. log(sum_j q_j) - log q_y -- mean over the batch.  z: the logits; returns (loss, dL/dz)."""
    z = np.asarray(z, dtype=np.float64)
    p = softmax(z)
    q = np.clip(p, eps, 1.0 - eps)
    B = z.shape[0]
    idx = np.arange(B)
    loss = float(np.mean(np.log(q.sum(axis=1)) - np.log(q[idx, y])))
    opened = (p > eps) & (p < 1.0 - eps)                     # clip_by_value passes the gradient strictly inside only
    g = np.where(opened, 1.0 / q.sum(axis=1, keepdims=True), 0.0)
    gy = np.zeros_like(g)
    gy[idx, y] = 1.0 / q[idx, y]
    g = g - np.where(opened, gy, 0.0)
    dz = p * (g - (p * g).sum(axis=1, keepdims=True)) / B
    return loss, dz


# ------------------------------------------------------------------ a17 AP loss
def ap_theta(z, N):
    """lidbox/losses.py:42-49: reference directions are the first N one-hot axes
    (:20), so z . c_T = z[:, :N]; theta = acos(.)."""
    return np.arccos(np.asarray(z)[:, :N])


def ap_loss_per_example(y_true, z, N, delta_weight=1.0):
    """lidbox/losses.py:25-40: L_l = sum_{l' != l} sigmoid(delta*(theta_l - theta_l'))."""
    th = ap_theta(z, N)
    th_l = th[np.arange(len(y_true)), y_true]
    sig = 1.0 / (1.0 + np.exp(-delta_weight * (th_l[:, None] - th)))
    mask = 1.0 - np.eye(N)[y_true]
    return (mask * sig).sum(axis=1)


def ap_loss(y_true, z, N, delta_weight=1.0):
    """Keras Loss.__call__ default reduction: mean over the batch."""
    return float(ap_loss_per_example(y_true, z, N, delta_weight).mean())


def ap_loss_grad(y_true, z, N, delta_weight=1.0, clamp=1e-6):
    """d mean(L) / d z, [B,D].  d acos(x)/dx = -1/sqrt(1-x^2); the denominator is
    floored at `clamp` (the build's documented choice where TF would return inf)."""
    z = np.asarray(z)
    B, D = z.shape
    x = z[:, :N]
    th = np.arccos(x)
    idx = np.arange(B)
    d = th[idx, y_true][:, None] - th
    s = 1.0 / (1.0 + np.exp(-delta_weight * d))
    ds = delta_weight * s * (1.0 - s) * (1.0 - np.eye(N)[y_true])
    dth = -ds
    dth[idx, y_true] += ds.sum(axis=1)
    dacos = -1.0 / np.sqrt(np.maximum(1.0 - x * x, clamp))
    g = np.zeros_like(z)
    g[:, :N] = dth * dacos / B
    return g


def l2_normalize(x, axis=1, eps=1e-12):
    """tf.math.l2_normalize: x * rsqrt(max(sum(x^2), eps))."""
    return x / np.sqrt(np.maximum((x * x).sum(axis=axis, keepdims=True), eps))


# ------------------------------------------------------------------ a18 C_avg
class AverageDetectionCost:
    """lidbox/metrics.py:6-111."""

    def __init__(self, N, thresholds, C_miss=1.0, C_fa=1.0, P_tar=0.5):
        assert N >= 2
        self.N = N
        self.th = np.asarray(thresholds, np.float32)
        assert self.th.ndim == 1
        self.C_miss, self.C_fa, self.P_tar = C_miss, C_fa, P_tar
        self.reset_states()

    def reset_states(self):                                                      # :46-48
        Th = len(self.th)
        self.fn = np.zeros((self.N, Th), np.float32)
        self.tp = np.zeros((self.N, Th), np.float32)
        self.fp_pairs = np.zeros((self.N, self.N, Th), np.float32)
        self.tn_pairs = np.zeros((self.N, self.N, Th), np.float32)

    def update_state(self, true_onehot, scores):                                 # :51-71
        true_onehot = np.asarray(true_onehot, np.float32)
        labels = true_onehot.argmax(axis=-1)
        tpos = true_onehot[:, :, None]
        tneg = (~tpos.astype(bool)).astype(np.float32)
        s = np.asarray(scores, np.float32)[:, :, None]
        pp = (s >= self.th).astype(np.float32)
        pn = (s < self.th).astype(np.float32)
        self.tp += (pp * tpos).sum(axis=0)
        self.fn += (pn * tpos).sum(axis=0)
        np.add.at(self.fp_pairs, labels, pp * tneg)
        np.add.at(self.tn_pairs, labels, pn * tneg)

    def per_threshold(self):                                                     # :73-99
        from .features_np import divide_no_nan
        P_miss = divide_no_nan(self.fn, self.fn + self.tp).mean(axis=0)
        inner = divide_no_nan(self.fp_pairs, self.fp_pairs + self.tn_pairs).sum(axis=1)
        P_fa = divide_no_nan(inner, np.float32(self.N - 1)).mean(axis=0)
        C = self.C_miss * self.P_tar * P_miss + self.C_fa * (1 - self.P_tar) * P_fa
        return P_miss, P_fa, C

    def result(self):                                                            # :100-103
        return float(self.per_threshold()[2].min())


class SparseAverageDetectionCost(AverageDetectionCost):
    """lidbox/metrics.py:114-119."""

    def update_state(self, labels, scores):
        super().update_state(np.eye(self.N, dtype=np.float32)[np.asarray(labels, np.int64)], scores)


# ------------------------------------------------------------------ 8f.1 xvector_2d: Conv2D over frequency + BatchNorm
FRAMES_2D = [("frame2d_1", 256, 5, 1), ("frame2d_2", 128, 3, 2), ("frame2d_3", 64, 3, 3),
             ("frame2d_4", 32, 3, 3)]           # (name, filters, kernel width, stride) -- xvector_2d.py:70-73, kernel (1, w), strides (1, s)
BN_MOMENTUM, BN_EPSILON = 0.99, 1e-3            # tf.keras.layers.BatchNormalization defaults (xvector_2d.py:36)


def conv_freq_out_len(F, k, s):
    """Keras Conv2D(padding="valid") along the frequency axis"""
    return (F - k) // s + 1 if F >= k else 0


def conv_freq_fwd(x, W, b, s, relu=True):
    """Conv2D(filters, (1, k), strides=(1, s), padding="valid", activation="relu") on x [B, T, F, C_in]
    (xvector_2d.py:29-35): every frame is convolved along frequency only.  W [1, k, C_in, C_out] (Keras kernel layout)."""
    _, k, Ci, Co = W.shape
    B, T, F, _ = x.shape
    Fo = conv_freq_out_len(F, k, s)
    idx = np.arange(Fo)[:, None] * s + np.arange(k)[None, :]
    col = x[:, :, idx, :].reshape(B, T, Fo, k * Ci)
    y = col @ W.reshape(k * Ci, Co) + b
    return np.maximum(y, 0) if relu else y


def batchnorm_fwd(x, gamma, beta, moving_mean, moving_var, training, momentum=BN_MOMENTUM, eps=BN_EPSILON):
    """BatchNormalization(axis=-1) on the 4-D conv output of xvector_2d.py:36 (tf.keras' fused path).  training: batch mean /
    population variance over every axis but the last for the normalisation; the moving mean moves by (1 - momentum) towards
    the batch mean, the moving variance towards the BESSEL-CORRECTED batch variance var * n / (n - 1) (the fused kernel's
    estimate, which Keras keeps: `_bessels_correction_test_only` defaults to True); inference: the moving statistics.
    Returns (y, moving_mean, moving_var)."""
    axes = tuple(range(x.ndim - 1))
    if training:
        mean = x.mean(axis=axes)
        var = ((x - mean) ** 2).mean(axis=axes)
        n = x.size // x.shape[-1]
        moving_mean = moving_mean * momentum + mean * (1 - momentum)
        moving_var = moving_var * momentum + var * (n / (n - 1) if n > 1 else 1.0) * (1 - momentum)
    else:
        mean, var = moving_mean, moving_var
    return gamma * (x - mean) / np.sqrt(var + eps) + beta, moving_mean, moving_var


def xvector_2d_freq_dims(F):
    dims = [F]
    for _, _, k, s in FRAMES_2D:
        dims.append(conv_freq_out_len(dims[-1], k, s))
    return dims


def xvector_2d_init(input_dim, num_outputs, seed=0, dtype=np.float32):
    """xvector_2d.py:66-93 in Keras layouts: Conv2D kernels [1, k, C_in, C_out] (glorot_uniform), biases 0, BatchNorm
    gamma 1 / beta 0 / moving_mean 0 / moving_variance 1, then the x-vector TDNN on cols * 32 input channels."""
    rng = np.random.default_rng(seed)
    p, c = {}, 1
    for name, f, k, s in FRAMES_2D:
        p[name + "_conv.W"] = glorot_uniform(rng, (1, k, c, f), k * c, k * f, dtype)
        p[name + "_conv.b"] = np.zeros(f, dtype)
        p[name + "_bn.gamma"] = np.ones(f, dtype)
        p[name + "_bn.beta"] = np.zeros(f, dtype)
        p[name + "_bn.moving_mean"] = np.zeros(f, dtype)
        p[name + "_bn.moving_variance"] = np.ones(f, dtype)
        c = f
    cols = xvector_2d_freq_dims(input_dim)[-1]
    if cols < 1:
        raise ValueError("input has too few frequency channels for the 2-D front-end")
    c = cols * c
    for name, f, k, s in XVECTOR_FRAMES:
        p[name + ".W"] = glorot_uniform(rng, (k, c, f), k * c, k * f, dtype)
        p[name + ".b"] = np.zeros(f, dtype)
        c = f
    c = 2 * c
    for name, u in XVECTOR_SEGMENTS + [("output", num_outputs)]:
        p[name + ".W"] = glorot_uniform(rng, (c, u), c, u, dtype)
        p[name + ".b"] = np.zeros(u, dtype)
        c = u
    return p


def xvector_2d_fwd(p, x, training=False, embedding=False):
    """xvector_2d.py:66-93 with output_activation="log_softmax".  x [B, T, F] -> (log-probs [B, N], updated moving
    statistics {name: value}); training selects batch statistics in the BatchNormalization layers."""
    B, T, F = x.shape
    h = x.reshape(B, T, F, 1)                                                     # reshape_to_image
    new_stats = {}
    for name, f, k, s in FRAMES_2D:
        h = conv_freq_fwd(h, p[name + "_conv.W"], p[name + "_conv.b"], s, relu=True)
        h, mm, mv = batchnorm_fwd(h, p[name + "_bn.gamma"], p[name + "_bn.beta"], p[name + "_bn.moving_mean"],
                                  p[name + "_bn.moving_variance"], training)
        new_stats[name + "_bn.moving_mean"], new_stats[name + "_bn.moving_variance"] = mm, mv
    h = h.reshape(B, T, h.shape[2] * h.shape[3])                                  # flatten_channels
    for name, f, k, s in XVECTOR_FRAMES:
        h = conv1d_causal_fwd(h, p[name + ".W"], p[name + ".b"], s, relu=True)
    pooled = stats_pool_fwd(h)
    if embedding:
        return dense_fwd(pooled, p["segment1.W"], p["segment1.b"], relu=False), new_stats
    s1 = dense_fwd(pooled, p["segment1.W"], p["segment1.b"])
    s2 = dense_fwd(s1, p["segment2.W"], p["segment2.b"])
    return log_softmax(dense_fwd(s2, p["output.W"], p["output.b"], relu=False)), new_stats
