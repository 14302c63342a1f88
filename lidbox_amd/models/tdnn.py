"""
Sequential TDNN engine: the host-side orchestration shared by `lidbox_amd.models.xvector` and
`lidbox_amd.models.cnn`.

A model is  Conv1D(causal, strided)+ReLU x n  ->  global pooling over time  ->  Dense x m  ->
log_softmax -- the structure of reference lidbox/models/xvector.py:46-67 and
lidbox/models/cnn.py:25-45.  Everything numeric is a liblidbox_hip.so call on preallocated
device buffers (no allocation, no host sync inside a step), so a whole train step can be
captured into a hipGraph.

HBM layout
  * parameters, gradients and Adam moments: ONE flat fp32 buffer each, per-layer views in
    Keras layouts (Conv1D kernel [k, C_in, C_out], Dense [in, out], bias [C_out]); every view
    starts on a 16-byte boundary.  The flat gradient buffer is what RCCL all-reduces.
  * activations: act[i] = [B, pad_i + T_i, C_i], pad_i = (k-1) leading zero rows of the layer
    that READS it.  A causal window is then k*C_i contiguous floats (row stride s*C_i): Conv1D
    forward / wgrad are GEMMs over implicit rows and no im2col buffer exists.  Gradient buffers
    d_act[i] share the layout (the pad rows just absorb unused dgrad output).
"""
import ctypes
import os
import math

import numpy as np
import torch

from .. import _native as nv


class ConvSpec:
    """Keras Conv1D(filters, kernel_size, strides, padding, dilation_rate).  The x-vector family only ever sets strides
    (xvector.py:38-39) with padding="causal"; cnn.create passes its `padding` argument through (cnn.py:25,33-36: "causal",
    "valid" or "same"); dilation_rate is this build's opt-in and, as in Keras, excludes strides > 1.  A dilated
    layer runs as k accumulating single-tap GEMMs over row-shifted views."""

    PADDINGS = ("causal", "valid", "same")

    def __init__(self, name, filters, kernel_size, strides, relu=True, dilation_rate=1, dense_kernel=False, padding="causal"):
        self.name, self.filters, self.k, self.s, self.relu = name, int(filters), int(kernel_size), int(strides), relu
        self.padding = str(padding).lower()
        if self.padding not in self.PADDINGS:
            raise ValueError("padding must be one of %s, got %r" % (self.PADDINGS, padding))
        # a Keras Dense applied to [B, T, C] is a pointwise conv; dense_kernel keeps its kernel shape [in, out]
        self.dense_kernel = bool(dense_kernel) and int(kernel_size) == 1
        self.d = int(dilation_rate) if int(kernel_size) > 1 else 1      # a single tap has nothing to dilate
        if self.d < 1 or self.k < 1 or self.s < 1:
            raise ValueError("kernel_size, strides and dilation_rate must be >= 1")
        if self.d > 1 and self.s > 1:
            raise ValueError("strides > 1 not supported in conjunction with dilation_rate > 1 (Keras Conv1D)")

    @property
    def pad(self):
        """causal left padding in rows"""
        return (self.k - 1) * self.d

    def geometry(self, T):
        """(zero rows ahead of the T input frames, zero rows behind them, output frames): output t reads the padded rows
        t*s + j*d, j < k.  TensorFlow's conventions: causal = k_eff - 1 rows ahead, then VALID; valid = no padding,
        floor((T - k_eff) / s) + 1 outputs; same = ceil(T / s) outputs, the missing rows split with the odd one behind."""
        ke = (self.k - 1) * self.d + 1
        if T <= 0:
            return (ke - 1 if self.padding == "causal" else 0), 0, 0
        if self.padding == "causal":
            return ke - 1, 0, (T - 1) // self.s + 1
        if self.padding == "valid":
            return 0, 0, ((T - ke) // self.s + 1 if T >= ke else 0)
        To = -(-T // self.s)
        total = max((To - 1) * self.s + ke - T, 0)
        return total // 2, total - total // 2, To


class FreqAttentionSpec:
    """clstm.frequency_attention(H, d_a, d_f) (reference clstm.py:31-42) between the last frame layer and the
    pooling: Dense(d_a, relu, no bias) "Wf_1" -> Dense(d_f, softmax, no bias) "Wf_2" -> bin-wise scaling."""

    def __init__(self, d_a=64, d_f=16):
        self.d_a, self.d_f = int(d_a), int(d_f)
        if not 1 <= self.d_f <= 64:
            raise ValueError("frequency attention supports 1 <= d_f <= 64 bins")


class FreqConvSpec:
    """FrameLayer2D of reference xvector_2d.py:26-46: Conv2D(filters, (1, k), strides=(1, s), padding="valid",
    activation="relu") along the FREQUENCY axis of [B, T, F, C] followed by BatchNormalization (Keras defaults: momentum
    0.99, epsilon 1e-3).  Every frame is an independent row batch, so the convolution is the same implicit-row GEMM as a
    Conv1D with "batch" = B*T and no padding."""

    def __init__(self, name, filters, kernel_width, stride, momentum=0.99, epsilon=1e-3, dropout_rate=None):
        self.name, self.filters, self.k, self.s = name, int(filters), int(kernel_width), int(stride)
        self.momentum, self.epsilon = float(momentum), float(epsilon)
        # Keras Dropout behind the BatchNormalization (xvector_2d.py:37-39,45-46); the reference's create() never sets it
        self.dropout_rate = float(dropout_rate or 0.0)
        if not 0.0 <= self.dropout_rate < 1.0:
            raise ValueError("dropout_rate must be in [0, 1)")
        if self.k < 1 or self.s < 1 or self.filters < 1:
            raise ValueError("filters, kernel width and stride must be >= 1")


def freq_out_len(F, k, s):
    """Keras Conv2D(padding="valid") along frequency"""
    return (F - k) // s + 1 if F >= k else 0


class DenseSpec:
    def __init__(self, name, units, relu=True):
        self.name, self.units, self.relu = name, int(units), relu


def conv_out_len(T, s):
    """Keras Conv1D(padding="causal"): left-pad k-1, VALID, stride s."""
    return (T - 1) // s + 1 if T > 0 else 0


def _align4(n):
    return (n + 3) & ~3


def _rows(t_ptr, batch_stride, row_stride, batch, rpb):
    return nv.Rows(t_ptr, int(batch_stride), int(row_stride), int(batch), int(rpb))


class _GemmFamily:
    """The C-ABI entry points of one GEMM family: fp32 MFMA (gemm.hip) or bf16 MFMA with fp32 storage and
    accumulation (gemm_bf16.hip, BASELINE config 5).  Same signatures, separate workspace sizing.  Entry points
    are looked up on nv.lib at call time (profiling tools wrap them there)."""

    _PREFIX = {"float32": "lidbox_gemm_", "bfloat16": "lidbox_gemm_bf16_"}

    def __init__(self, compute_dtype):
        if compute_dtype in ("float32", "fp32", "f32", torch.float32):
            self.name = "float32"
        elif compute_dtype in ("bfloat16", "bf16", torch.bfloat16):
            self.name = "bfloat16"
        else:
            raise ValueError("compute_dtype must be 'float32' or 'bfloat16', got %r" % (compute_dtype,))
        self._prefix = self._PREFIX[self.name]

    def __getattr__(self, op):          # nn, nt, tn, rows_workspace, tn_workspace
        if op.startswith("_"):
            raise AttributeError(op)
        return getattr(nv.lib, self._prefix + op)


class _Workspace:
    """All per-(B, T) device buffers of one model."""

    def __init__(self, model, B, T):
        dev = model.device
        self.B, self.T = B, T
        f32 = dict(dtype=torch.float32, device=dev)
        convs = model.convs
        self.Ts = [T]
        self.pads, self.rpads = [], []       # zero rows ahead of / behind the frames of act[i], for the conv that reads it
        for c in convs:
            pl, pr, To = c.geometry(self.Ts[-1])
            self.pads.append(pl)
            self.rpads.append(pr)
            self.Ts.append(To)
        self.pads.append(0)
        self.rpads.append(0)
        chans = [model.input_dim] + [c.filters for c in convs]
        # zero rows BEHIND an utterance's frames (bf16-storage path only): the output-stationary dgrad of a conv with k > s
        # reads its output gradient through windows that run up to `trail` rows past the last frame (SequentialTDNN._dgrad_residues)
        self.trail = [0] * len(chans)
        if model.bf16_storage:
            for i in range(1, len(convs) - 1):               # not the last conv: the pooling reads act[-1] as [B, T, C]
                self.trail[i + 1] = model._dgrad_trail(i, self.Ts[i])
        # activations (zero-initialised once: the pad / trail rows stay zero forever)
        self.act = [torch.zeros((B, self.pads[i] + self.Ts[i] + self.rpads[i] + self.trail[i], chans[i]), **f32) for i in range(len(chans))]
        self.dact = [None] + [torch.zeros_like(a) for a in self.act[1:]]
        # bf16-storage GEMMs (compute_dtype "bfloat16"): bf16 shadows of the conv inputs (A of forward) and of the conv output
        # gradients (A of dgrad), same element layout as the fp32 buffers; written by the producing GEMM's epilogue.
        # A layer takes the shadow path when its operand strides / K are multiples of 8 bf16 elements.
        self.act16 = [None] * len(self.act)
        self.dact16 = [None] * len(self.act)
        self.d16_fresh = set()               # indices j whose dact16[j] currently holds bf16(dact[j]) (reset by every backward pass)
        if model.bf16_storage:
            for i, c in enumerate(convs):
                if model.shadow_fwd_ok(i):
                    self.act16[i] = torch.zeros(self.act[i].shape, dtype=torch.bfloat16, device=dev)
                if (i >= 1 and model.shadow_dgrad_ok(i)) or model.shadow_wgrad_ok(i):
                    # rows padded to 8 elements when the channel count is not a multiple of 8 (frame5's 1500): wgrad reads
                    # whole 16-byte pieces; the pad columns stay zero
                    shp = self.act[i + 1].shape
                    self.dact16[i + 1] = torch.zeros((shp[0], shp[1], (shp[2] + 7) // 8 * 8), dtype=torch.bfloat16, device=dev)
        # all-shadow mode with statistics pooling over short utterances: the last conv writes ONLY a bf16 shadow of its output
        # (rows padded to 8 channels) and both pooling passes read that (lidbox_stats_pool_*_bf16): no fp32 copy of the widest
        # activation is written or read.  LIDBOX_BF16_POOL_FP32=1 keeps the fp32 copy (tests, A/B).
        self.last16 = None
        if (model.bf16_storage and model.bf16_only and model.pool == "stats" and model.attention is None and 1 <= self.Ts[-1] <= 40
                and chans[-1] % 4 == 0 and model.shadow_fwd_ok(len(convs) - 1) and convs[-1].d == 1
                and os.environ.get("LIDBOX_BF16_POOL_FP32", "0") != "1"):
            self.last16 = torch.zeros((B, self.Ts[-1], (chans[-1] + 7) // 8 * 8), dtype=torch.bfloat16, device=dev)
            # the fp32 buffers of the last layer stay allocated (the row descriptors are derived from them) but nothing writes
            # them in this mode: poisoned, so that a reader that should have taken the shadow fails loudly instead of seeing zeros
            self.act[-1].fill_(float("nan"))
            self.dact[-1].fill_(float("nan"))
        fe = model.frontend
        if fe:
            # 2-D front-end (xvector_2d.py:69-73): model input [B, T, F]; layer i: a = relu(conv) [B*T*F_i+1, C_i+1] dense,
            # y = BatchNorm(a) dense -- except the last layer's, which IS the first Conv1D's input (act[0])
            self.dact[0] = torch.zeros_like(self.act[0])
            self.fe_in = torch.zeros((B, T, model.model_input_dim), **f32)
            self.fe_R = [B * T * model.fe_dims[i + 1] for i in range(len(fe))]
            self.fe_a = [torch.zeros((self.fe_R[i], l.filters), **f32) for i, l in enumerate(fe)]
            self.fe_dz = [torch.zeros_like(a) for a in self.fe_a]
            self.fe_y = [torch.zeros_like(a) for a in self.fe_a[:-1]]
            self.fe_dy = [torch.zeros_like(a) for a in self.fe_a[:-1]]
            self.fe_consts = [torch.zeros((4, l.filters), **f32) for l in fe]        # mean, invstd, scale, shift
            bn_bytes = max(int(nv.lib.lidbox_bn_workspace(self.fe_R[i], l.filters)) for i, l in enumerate(fe))
            self.bn_ws = torch.empty(max(16, bn_bytes), dtype=torch.uint8, device=dev)
        C_last = chans[-1]
        att = model.attention
        if att is not None:
            n = B * self.Ts[-1]
            self.fa_x1 = torch.zeros((n, att.d_a), **f32)            # relu(H Wf_1)
            self.fa_F = torch.zeros((n, att.d_f), **f32)             # logits, then softmax in place
            self.hw = torch.zeros_like(self.act[-1])                 # attention output = pooling input
            self.d_hw = torch.zeros_like(self.act[-1])
            self.d_logit = torch.zeros((n, att.d_f), **f32)
            self.d_x1 = torch.zeros((n, att.d_a), **f32)
        P = 2 * C_last if model.pool == "stats" else C_last
        self.pooled = torch.zeros((B, P), **f32)
        self.dpooled = torch.zeros((B, P), **f32)
        self.h = [torch.zeros((B, d.units), **f32) for d in model.denses]          # last = logits z
        self.dh = [torch.zeros((B, d.units), **f32) for d in model.denses]
        self.logp = torch.zeros((B, model.denses[-1].units), **f32)
        self.emb = torch.zeros((B, model.denses[0].units), **f32)
        self.loss = torch.zeros(4, **f32)
        # one GEMM workspace sized for the largest split (wgrad partials, small-M split-K partials)
        ws_bytes = 16
        g = model.gemm
        if fe:
            cin_f = 1
            for i, l in enumerate(fe):
                gf = model.fe_gemm(i)
                M, kk = self.fe_R[i], l.k * cin_f
                if M > 0:
                    ws_bytes = max(ws_bytes, gf.tn_workspace(M, kk, l.filters), gf.rows_workspace(M, l.filters, kk),
                                   gf.rows_workspace(M, kk, l.filters))
                cin_f = l.filters
        cin = model.input_dim
        for i, c in enumerate(convs):
            M = B * self.Ts[i + 1]
            if M > 0:
                kk = cin if c.d > 1 else c.k * cin               # dilated layers run one tap per GEMM
                ws_bytes = max(ws_bytes, g.tn_workspace(M, kk, c.filters),
                               g.rows_workspace(M, c.filters, kk), g.rows_workspace(M, kk, c.filters))
                if model.bf16_storage and model.shadow_wgrad_ok(i):
                    ws_bytes = max(ws_bytes, nv.lib.lidbox_gemm_bf16s_tn_workspace(M, kk, c.filters))
            cin = c.filters
        if att is not None and B * self.Ts[-1] > 0:
            n = B * self.Ts[-1]
            for (kk, nn_) in ((C_last, att.d_a), (att.d_a, att.d_f)):
                ws_bytes = max(ws_bytes, g.tn_workspace(n, kk, nn_), g.rows_workspace(n, nn_, kk), g.rows_workspace(n, kk, nn_))
        din = P
        for d in model.denses:
            for gd in {g, model.dense_gemm}:
                ws_bytes = max(ws_bytes, gd.tn_workspace(B, din, d.units), gd.rows_workspace(B, d.units, din),
                               gd.rows_workspace(B, din, d.units))
            din = d.units
        self.gemm_ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        self.gemm_ws2 = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)     # wgrad side stream's own workspace
        # carried reduces (lidbox_hip.h: lidbox_reduce_job_t): a wgrad's slices wait in their workspace until a later GEMM
        # launch has summed them, so consecutive wgrads alternate between two regions.  Both are buffers of their own: gemm_ws2
        # belongs to the wgrad side streams (`_launch_wgrad`), which nothing orders against the main stream's carried slices
        # until `join_wgrad` at the end of a stage
        self.tn_regions = (torch.empty(ws_bytes, dtype=torch.uint8, device=dev), torch.empty(ws_bytes, dtype=torch.uint8, device=dev))
        self.pending = []                    # [(nv.ReduceJob, region index)], oldest first

    def select_input_buffer(self, parity):
        """Double-buffered model input (Trainer's feature prefetch: the features of batch t + 1 are extracted beside the
        GEMMs of step t, the counterpart of the tf.data prefetch around lidbox/data/steps.py:725-736): makes act[0] (and its
        bf16 shadow) the buffer `parity`; the second one is allocated on first use.  Every descriptor of the first layer is
        derived from act[0] when a step is issued / captured."""
        if hasattr(self, "fe_in"):
            raise ValueError("no double-buffered input behind a 2-D front-end")
        if not hasattr(self, "_in_bufs"):
            self._in_bufs = [(self.act[0], self.act16[0]), None]
        if self._in_bufs[parity] is None:
            a16 = self.act16[0]
            self._in_bufs[parity] = (torch.zeros_like(self.act[0]), None if a16 is None else torch.zeros_like(a16))
        self.act[0], self.act16[0] = self._in_bufs[parity]

    def input_view(self):
        """[B, T, C0] view of where the model input lives: act[0] behind its causal zero rows, or the 2-D front-end's
        input buffer."""
        if hasattr(self, "fe_in"):
            return self.fe_in
        return self.act[0][:, self.pads[0]:, :]

    def input_target(self):
        """(pointer, floats between utterances, T, C) of the model input buffer (what Trainer / _load_input fill)"""
        v = self.input_view()
        return ctypes.c_void_p(v.data_ptr()), v.stride(0), v.shape[1], v.shape[2]


class SequentialTDNN:
    """convs -> pool -> denses -> log_softmax, parameters in one flat buffer."""

    def __init__(self, input_shape, convs, pool, denses, name="tdnn", output_activation="log_softmax",
                 channel_dropout_rate=0.0, seed=None, device=None, compute_dtype="float32", attention=None, frontend=None):
        if not torch.cuda.is_available():
            raise nv.LidboxHipError("lidbox_amd models need a HIP device (no CPU fallback)")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.name = name
        self.input_shape = tuple(input_shape)
        self.model_input_dim = int(input_shape[-1])            # channels of what the caller passes
        self.frontend = list(frontend) if frontend else None
        self.input_dim = self.model_input_dim                  # channels of the first Conv1D's input
        if self.frontend:
            # [B, T, F] -> reshape_to_image -> FrameLayer2D x n -> flatten_channels (xvector_2d.py:68-76)
            self.fe_dims = [self.model_input_dim]
            for l in self.frontend:
                self.fe_dims.append(freq_out_len(self.fe_dims[-1], l.k, l.s))
            if self.fe_dims[-1] < 1:
                raise ValueError("input has too few frequency channels (%d) for the 2-D front-end" % self.model_input_dim)
            self.input_dim = self.fe_dims[-1] * self.frontend[-1].filters
        self.convs, self.pool, self.denses = list(convs), pool, list(denses)
        self.attention = attention
        assert pool in ("stats", "avg")
        if attention is not None and self.convs[-1].filters % attention.d_f:
            # clstm.py:32
            raise ValueError("amount of frequency channels (%d) must be evenly divisible by the amount of frequency "
                             "attention bins (d_f=%d)" % (self.convs[-1].filters, attention.d_f))
        # the reference takes getattr(tf.nn, output_activation) (cnn.py:43-44, xvector_2d.py:86-87): log_softmax (its default),
        # softmax, or no activation ("linear" / None)
        if output_activation == "linear":
            output_activation = None
        if output_activation not in ("log_softmax", "softmax", None):
            raise ValueError("output_activation must be 'log_softmax', 'softmax' or None / 'linear'")
        self.output_activation = output_activation
        self.channel_dropout_rate = float(channel_dropout_rate)
        if not 0.0 <= self.channel_dropout_rate < 1.0:
            raise ValueError("channel_dropout_rate must be in [0, 1)")
        self.dropout_seed = int(np.random.default_rng(seed).integers(1, 2 ** 62))
        self._dropout_calls = 0
        self.embedding_layer = self.denses[0].name
        # GEMM arithmetic: "float32" (exact fp32 MFMA) or "bfloat16" (operands rounded to bf16 on chip, fp32
        # accumulate; all buffers, the master weights and every non-GEMM kernel stay fp32)
        self.gemm = _GemmFamily(compute_dtype)
        self.compute_dtype = self.gemm.name
        # The dense head (M = batch rows: three layers, < 0.5 % of the flops) stays on the fp32 family under bf16 compute:
        # its launches are latency-bound and the fp32 family's 64 x 64 tiles + tuned splits run them faster than the bf16
        # family's 128 x 128 tile (0.910 -> 0.871 ms/step at bs 256, neutral at bs 512).  LIDBOX_BF16_DENSE=1
        # puts the head on the bf16 family as well (A/B aid).
        self.dense_gemm = self.gemm if os.environ.get("LIDBOX_BF16_DENSE") == "1" else _GemmFamily("float32")
        if self.compute_dtype == "bfloat16":
            widths = [self.input_dim] + [c.filters for c in self.convs] + [d.units for d in self.denses]
            if attention is not None:
                widths += [attention.d_a, attention.d_f]
            if any(w % 4 for w in widths):
                raise ValueError("bfloat16 compute needs channel / unit counts that are multiples of 4, got %s" % widths)
        # ---- flat parameter layout
        self.layout = {}           # name -> (offset, shape)
        off = 0
        if self.frontend:          # first in the flat buffer: their gradients complete last, in the lowest all-reduce bucket
            cin = 1
            for l in self.frontend:
                self.layout[l.name + "_conv.W"] = (off, (1, l.k, cin, l.filters)); off = _align4(off + l.k * cin * l.filters)
                for suffix in ("_conv.b", "_bn.gamma", "_bn.beta"):
                    self.layout[l.name + suffix] = (off, (l.filters,)); off = _align4(off + l.filters)
                cin = l.filters
        cin = self.input_dim
        for c in self.convs:
            self.layout[c.name + ".W"] = (off, (cin, c.filters) if c.dense_kernel else (c.k, cin, c.filters))
            off = _align4(off + c.k * cin * c.filters)
            self.layout[c.name + ".b"] = (off, (c.filters,)); off = _align4(off + c.filters)
            cin = c.filters
        if attention is not None:                      # bias-free Dense kernels, Keras names Wf_1 / Wf_2 (clstm.py:35-36)
            self.layout["Wf_1.W"] = (off, (cin, attention.d_a)); off = _align4(off + cin * attention.d_a)
            self.layout["Wf_2.W"] = (off, (attention.d_a, attention.d_f)); off = _align4(off + attention.d_a * attention.d_f)
        din = 2 * cin if pool == "stats" else cin
        for d in self.denses:
            self.layout[d.name + ".W"] = (off, (din, d.units)); off = _align4(off + din * d.units)
            self.layout[d.name + ".b"] = (off, (d.units,)); off = _align4(off + d.units)
            din = d.units
        self.num_flat = off
        self.flat = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.flat_grad = torch.zeros_like(self.flat)
        # non-trainable state (BatchNormalization moving statistics): its own flat buffer, not touched by Adam
        self.state_layout, soff = {}, 0
        for l in self.frontend or []:
            for suffix in ("_bn.moving_mean", "_bn.moving_variance"):
                self.state_layout[l.name + suffix] = (soff, (l.filters,)); soff = _align4(soff + l.filters)
        self.state = torch.zeros(max(soff, 4), dtype=torch.float32, device=self.device)
        self._init_weights(seed)
        # bf16-storage GEMM path (LIDBOX_BF16_STORAGE=0 keeps the fp32-source bf16 kernels: A/B aid): bf16 weight shadows,
        # refreshed from the fp32 master copy at the start of every forward pass -- `flat16` has the element layout of `flat`
        # (dgrad reads a Keras kernel [k*C_in, C_out] as the [N][K] operand it is) but only the kernels listed in
        # `_flat16_live` are refreshed (`_refresh_bf16_weights`); every other element stays zero and `_p16` refuses to hand
        # it out.  `w16t[i]` is conv i's kernel transposed to [C_out, k*C_in] (forward's [N][K] operand)
        # (the storage path's output-stationary dgrad and its trail rows are laid out for causal windows)
        self.bf16_storage = self.compute_dtype == "bfloat16" and os.environ.get("LIDBOX_BF16_STORAGE", "1") != "0" \
            and attention is None and not self.frontend and all(c.padding == "causal" for c in self.convs)
        if self.bf16_storage and not any(self.shadow_fwd_ok(i) for i in range(len(self.convs))):
            self.bf16_storage = False                    # no layer qualifies (e.g. the CNN's 12- / 500-channel layers): nothing to shadow
        if self.bf16_storage:
            self.flat16 = torch.zeros(self.num_flat, dtype=torch.bfloat16, device=self.device)
            cin = self.input_dim
            self.w16t = []
            # wd16[i]: dgrad's weight images of conv i where flat16 itself cannot serve -- {"fused": [k*C_in][C_out padded to
            # 8]} for k <= s with a channel count that is not a multiple of 8 (frame5's 1500), {rho: [C_in][Q * C_out padded]}
            # per row residue for k > s (the taps rho + q*s side by side, last tap first; _dgrad_residues)
            self.wd16 = []
            # bf16_only: every conv runs forward, wgrad and dgrad on the shadows, so the fp32 copies of the intermediate
            # activations act[1..n-1] and of all conv output gradients dact[1..n] have no reader -- they are not written at
            # all (the ReLU masks come from the shadows' signs).  LIDBOX_BF16_FP32_COPIES=1 keeps writing them (tests, A/B).
            n = len(self.convs)
            self.bf16_only = os.environ.get("LIDBOX_BF16_FP32_COPIES", "0") != "1" and self.pool == "stats" and all(
                self.shadow_wgrad_ok(i) and (i == 0 or (self.shadow_dgrad_ok(i) and (self.convs[i].k <= self.convs[i].s or i < n - 1)))
                for i in range(n))
            bf = dict(dtype=torch.bfloat16, device=self.device)
            for i, c in enumerate(self.convs):
                self.w16t.append(torch.zeros((c.filters, c.k * cin), **bf) if self.shadow_fwd_ok(i) else None)
                img = {}
                if i >= 1 and self.shadow_dgrad_ok(i):
                    cp = (c.filters + 7) // 8 * 8
                    if c.k > c.s and i < len(self.convs) - 1:
                        for rho, Q, _, _ in self._dgrad_residues(i, 1 << 20):
                            img[rho] = torch.zeros((cin, Q * cp), **bf)
                    elif c.k <= c.s and cp != c.filters:
                        img["fused"] = torch.zeros((c.k * cin, cp), **bf)
                self.wd16.append(img)
                cin = c.filters
        if not self.bf16_storage:
            self.bf16_only = False
        self._ws = {}
        # optional second HIP stream: wgrad GEMMs run on it concurrently with the dgrad chain (they only
        # share read-only inputs), which fills the tail rounds and the nearly empty dense-layer launches
        self.wgrad_stream = None
        self.head_wgrad_stream = None        # dense-head wgrads only (see _launch_wgrad)

    # ------------------------------------------------------------------ parameters
    def _init_weights(self, seed):
        """Keras defaults: glorot_uniform kernels, zero biases."""
        rng = np.random.default_rng(seed)
        host = np.zeros(self.num_flat, np.float32)
        for name, (off, shape) in self.layout.items():
            if name.endswith(".W"):
                if len(shape) == 4:                          # Conv2D kernel [1, k, C_in, C_out]
                    _, k, ci, co = shape
                    fan_in, fan_out = k * ci, k * co
                elif len(shape) == 3:
                    k, ci, co = shape
                    fan_in, fan_out = k * ci, k * co
                else:
                    fan_in, fan_out = shape
                limit = math.sqrt(6.0 / (fan_in + fan_out))
                n = int(np.prod(shape))
                host[off:off + n] = rng.uniform(-limit, limit, size=n).astype(np.float32)
            elif name.endswith("_bn.gamma"):
                host[off:off + shape[0]] = 1.0
        self.flat.copy_(torch.from_numpy(host))
        for name, (off, shape) in self.state_layout.items():
            self.state[off:off + shape[0]] = 1.0 if name.endswith("moving_variance") else 0.0

    def param(self, name, grad=False):
        if name in self.state_layout:
            off, shape = self.state_layout[name]
            return self.state[off:off + int(np.prod(shape))].view(shape)
        off, shape = self.layout[name]
        buf = self.flat_grad if grad else self.flat
        return buf[off:off + int(np.prod(shape))].view(shape)

    def named_parameters(self):
        return [(n, self.param(n)) for n in self.layout]

    def count_params(self):
        """Keras `Model.count_params()`: trainable + non-trainable (BatchNormalization moving statistics)"""
        return sum(int(np.prod(s)) for _, s in list(self.layout.values()) + list(self.state_layout.values()))

    def get_weights(self):
        """dict name -> numpy array in Keras layouts (trainable parameters and the BatchNormalization moving statistics)."""
        return {n: self.param(n).detach().cpu().numpy().copy() for n in list(self.layout) + list(self.state_layout)}

    def set_weights(self, weights):
        for n, w in weights.items():
            self.param(n).copy_(torch.as_tensor(np.asarray(w, np.float32)).to(self.device).reshape(self.param(n).shape))

    def _cin(self, i):
        return self.input_dim if i == 0 else self.convs[i - 1].filters

    def shadow_fwd_ok(self, i):
        """forward of conv i can read bf16 shadows: rows of k*C_in bf16 at strides that are multiples of 8 elements"""
        c = self.convs[i]
        return c.d == 1 and self._cin(i) % 8 == 0

    def shadow_wgrad_ok(self, i):
        """wgrad of conv i can read bf16 shadows of its input windows and of its output gradient"""
        c = self.convs[i]
        return c.d == 1 and self._cin(i) % 8 == 0 and c.filters % 4 == 0

    def shadow_dgrad_ok(self, i):
        """dgrad of conv i can read the bf16 shadow of its output gradient (rows padded to 8 channels where needed)"""
        c = self.convs[i]
        return c.d == 1 and c.filters % 4 == 0 and self._cin(i) % 8 == 0

    def _dgrad_residues(self, i, T):
        """Output-stationary dgrad of conv i (d = 1) over inputs of T frames: the padded input row p = u*s + rho receives
        sum_q dY[u - q] . W[rho + q*s]^T over the taps j = rho + q*s < k, i.e. ONE GEMM per residue rho whose A rows are
        windows of Q consecutive output-gradient rows and whose B stacks those taps -- every row written exactly once, no
        accumulation passes, no zero fills.  Returns [(rho, Q, u_min, u_max)]: rows p >= pad only (the causal pad rows keep
        their zeros); windows start at dY row u - Q + 1 >= 0 and may run past the last one (see _dgrad_trail)."""
        c = self.convs[i]
        res = []
        for rho in range(min(c.s, c.k)):
            Q = (c.k - rho + c.s - 1) // c.s
            u_min = -((rho - c.pad) // c.s)                   # ceil((pad - rho) / s)
            u_max = (c.pad + T - 1 - rho) // c.s
            if T > 0 and u_max >= u_min:
                res.append((rho, Q, u_min, u_max))
        return res

    def _dgrad_trail(self, i, T):
        """zero rows needed behind the last output-gradient row of an utterance by conv i's output-stationary dgrad"""
        c = self.convs[i]
        if not (c.d == 1 and c.k > c.s and self.shadow_dgrad_ok(i)) or T <= 0:
            return 0
        To = conv_out_len(T, c.s)
        return max(0, max(u_max for _, _, _, u_max in self._dgrad_residues(i, T)) - (To - 1))

    def _refresh_bf16_weights(self):
        """the bf16 weight images from the fp32 master copy, one launch: the transposed conv kernels (forward), the dgrad
        images, and -- of flat16 -- only the kernels a dgrad reads in place (k <= s, channels a multiple of 8: frame3 / frame4
        of the x-vector); the rest of flat16 has no reader and is not refreshed (it stays zero)"""
        mats = getattr(self, "_w16t_descs", None)
        if mats is None:
            items = []
            live = self._flat16_live = set()          # parameters whose flat16 image is refreshed (what `_p16` may return)
            for i, c in enumerate(self.convs):
                off, cin, co = self.layout[c.name + ".W"][0], self._cin(i), c.filters
                if self.w16t[i] is not None:
                    items.append((off, c.k * cin, co, self.w16t[i].data_ptr(), c.k * cin, 1))
                if i >= 1 and self.shadow_dgrad_ok(i) and not self.wd16[i]:
                    items.append((off, c.k * cin, co, self.flat16.data_ptr() + 2 * off, co, 0))       # read through _p16
                    live.add(c.name + ".W")
                for key, img in self.wd16[i].items():
                    if key == "fused":
                        items.append((off, c.k * cin, co, img.data_ptr(), img.shape[1], 0))
                        continue
                    Q = (c.k - key + c.s - 1) // c.s
                    cp = img.shape[1] // Q
                    for q in range(Q):                        # tap rho + q*s sits in column block Q-1-q
                        items.append((off + (key + q * c.s) * cin * co, cin, co, img.data_ptr() + 2 * (Q - 1 - q) * cp,
                                      img.shape[1], 0))
            mats = (nv.WeightShadow * max(1, len(items)))(*[nv.WeightShadow(*it) for it in items])
            self._w16t_descs, self._w16t_n = mats, len(items)
        nv.check(nv.lib.lidbox_refresh_bf16_weights(nv.ptr(self.flat), None, self.num_flat, mats, self._w16t_n, nv.current_stream()))

    def _p16(self, name):
        """the bf16 image of parameter `name` inside flat16 -- only for the kernels `_refresh_bf16_weights` keeps current"""
        if name not in getattr(self, "_flat16_live", ()):
            raise RuntimeError("flat16 holds no refreshed image of %r (refreshed: %s)" % (name, sorted(getattr(self, "_flat16_live", ()))))
        off, _ = self.layout[name]
        return ctypes.c_void_p(self.flat16.data_ptr() + 2 * off)

    @staticmethod
    def _rows16(view_rows, t32, t16):
        """the rows descriptor `view_rows` of fp32 tensor t32 [B, R, C], re-based onto its bf16 shadow t16: same element
        offsets, or -- for a shadow whose rows are padded to C16 > C -- the same (row, column) positions"""
        off = (view_rows.base - t32.data_ptr()) // 4
        C, C16 = t32.shape[-1], t16.shape[-1]
        if C16 == C:
            return nv.Rows(t16.data_ptr() + 2 * off, view_rows.batch_stride, view_rows.row_stride, view_rows.batch,
                           view_rows.rows_per_batch)
        assert view_rows.batch_stride % C == 0 and view_rows.row_stride % C == 0
        r, col = divmod(off, C)
        return nv.Rows(t16.data_ptr() + 2 * (r * C16 + col), view_rows.batch_stride // C * C16, view_rows.row_stride // C * C16,
                       view_rows.batch, view_rows.rows_per_batch)

    def _sp(self, name):
        off, _ = self.state_layout[name]
        return ctypes.c_void_p(self.state.data_ptr() + 4 * off)

    def fe_gemm(self, i):
        """GEMM family of front-end layer i: the first one contracts over k = 5 single-channel taps (row stride 1 float),
        which the bf16 family's 16-byte operand rule excludes -- it always runs in the fp32 family"""
        return self.gemm if i > 0 else _GemmFamily("float32")

    def _p(self, name, grad=False):
        off, _ = self.layout[name]
        base = (self.flat_grad if grad else self.flat).data_ptr()
        return ctypes.c_void_p(base + 4 * off)

    # ------------------------------------------------------------------ workspace
    def workspace(self, B, T):
        key = (int(B), int(T))
        ws = self._ws.get(key)
        if ws is None:
            if len(self._ws) >= 4:                     # keep the cache small
                self._ws.pop(next(iter(self._ws)))
            ws = _Workspace(self, *key)
            self._ws[key] = ws
        return ws

    # ------------------------------------------------------------------ forward
    def _conv_rows_in(self, ws, i):
        a = ws.act[i]
        c = self.convs[i]
        Tp, C = a.shape[1], a.shape[2]
        return _rows(a.data_ptr(), Tp * C, c.s * C, ws.B, ws.Ts[i + 1])

    def _tap_rows(self, buf, ws, i, j):
        """rows of act[i] / dact[i] that tap j of dilated conv i touches: padded rows j*d + [0, T_out)."""
        c = self.convs[i]
        Tp, C = buf.shape[1], buf.shape[2]
        return _rows(buf.data_ptr() + 4 * j * c.d * C, Tp * C, C, ws.B, ws.Ts[i + 1])

    def _tap_weight(self, c, cin, j, grad=False):
        """[C_in, C_out] block of tap j inside the Keras kernel [k, C_in, C_out]"""
        return ctypes.c_void_p(self._p(c.name + ".W", grad).value + 4 * j * cin * c.filters)

    def _rows_out(self, buf, ws, i):
        """rows of act[i] / dact[i] behind the pad (what the producing layer writes)."""
        Tp, C = buf.shape[1], buf.shape[2]
        return _rows(buf.data_ptr() + 4 * ws.pads[i] * C, Tp * C, C, ws.B, ws.Ts[i])

    def _fe_rows_in(self, ws, i):
        """implicit rows of front-end layer i's input: one row per (frame, output column), k * C_in contiguous floats"""
        l = self.frontend[i]
        cin = 1 if i == 0 else self.frontend[i - 1].filters
        src = ws.fe_in if i == 0 else ws.fe_y[i - 1]
        return _rows(src.data_ptr(), self.fe_dims[i] * cin, l.s * cin, ws.B * ws.T, self.fe_dims[i + 1])

    def _fe_last_rows(self, buf, ws):
        """the last front-end layer's output rows (b, t, col) inside act[0] / dact[0] (flatten_channels, xvector_2d.py:76)"""
        Tp, C = buf.shape[1], buf.shape[2]
        return _rows(buf.data_ptr() + 4 * ws.pads[0] * C, Tp * C, self.frontend[-1].filters, ws.B, ws.T * self.fe_dims[-1])

    def _forward_frontend(self, ws, training, update_moving):
        st, lib = nv.current_stream(), nv.lib
        cin = 1
        for i, l in enumerate(self.frontend):
            R, Co, K = ws.fe_R[i], l.filters, l.k * cin
            if R == 0:
                break
            g = self.fe_gemm(i)
            nv.check(g.nn(self._fe_rows_in(ws, i), self._p(l.name + "_conv.W"), Co, _rows(ws.fe_a[i].data_ptr(), 0, Co, 1, R),
                          K, Co, nv.EPI_BIAS_RELU, self._p(l.name + "_conv.b"), nv.ptr(ws.gemm_ws), ws.gemm_ws.numel(), st))
            c = ws.fe_consts[i]
            cp = [ctypes.c_void_p(c.data_ptr() + 4 * j * Co) for j in range(4)]
            if training:
                mm = self._sp(l.name + "_bn.moving_mean") if update_moving else None
                mv = self._sp(l.name + "_bn.moving_variance") if update_moving else None
                nv.check(lib.lidbox_bn_train_stats(nv.ptr(ws.fe_a[i]), R, Co, self._p(l.name + "_bn.gamma"), self._p(l.name + "_bn.beta"),
                                                   l.epsilon, l.momentum, mm, mv, cp[0], cp[1], cp[2], cp[3],
                                                   nv.ptr(ws.bn_ws), ws.bn_ws.numel(), st))
            else:
                nv.check(lib.lidbox_bn_infer_consts(self._p(l.name + "_bn.gamma"), self._p(l.name + "_bn.beta"),
                                                    self._sp(l.name + "_bn.moving_mean"), self._sp(l.name + "_bn.moving_variance"),
                                                    l.epsilon, Co, cp[2], cp[3], st))
            last = i == len(self.frontend) - 1
            y = self._fe_last_rows(ws.act[0], ws) if last else _rows(ws.fe_y[i].data_ptr(), 0, Co, 1, R)
            nv.check(lib.lidbox_bn_apply(nv.ptr(ws.fe_a[i]), R, Co, cp[2], cp[3], y, st))
            if training and l.dropout_rate > 0:
                nv.check(lib.lidbox_dropout_rows(y, Co, l.dropout_rate, self._fe_dropout_seed(i), self._dropout_step_ptr(), st))
            cin = Co

    def _fe_dropout_seed(self, i):
        # dropout_seed_mix: the Trainer's per-rank offset under data parallelism (train.Trainer._rank_mix), 0 otherwise
        return (self.dropout_seed + getattr(self, "dropout_seed_mix", 0) + 0xD1B54A32D192ED03 * (i + 1)) & (2 ** 64 - 1)

    def _dropout_step_ptr(self):
        """device int64 that keys the FrameLayer2D dropout masks: the Trainer points it at its Adam step (a fresh mask per
        step, also under graph replay); standalone training-mode calls use a constant zero"""
        t = getattr(self, "dropout_step", None)
        return None if t is None else nv.ptr(t)

    def _backward_frontend(self, ws):
        """dact[0] holds d loss / d (front-end output).  BatchNorm + ReLU backward, wgrad / bias gradient, dgrad per layer."""
        st, lib = nv.current_stream(), nv.lib
        gws, gws_n = nv.ptr(ws.gemm_ws), ws.gemm_ws.numel()
        n = len(self.frontend)
        for i in range(n - 1, -1, -1):
            l = self.frontend[i]
            cin = 1 if i == 0 else self.frontend[i - 1].filters
            R, Co, K = ws.fe_R[i], l.filters, l.k * cin
            if R == 0:
                continue
            g = self.fe_gemm(i)
            c = ws.fe_consts[i]
            dy = self._fe_last_rows(ws.dact[0], ws) if i == n - 1 else _rows(ws.fe_dy[i].data_ptr(), 0, Co, 1, R)
            if l.dropout_rate > 0:               # the forward mask, regenerated on the gradient (same seed, same device step)
                nv.check(lib.lidbox_dropout_rows(dy, Co, l.dropout_rate, self._fe_dropout_seed(i), self._dropout_step_ptr(), st))
            nv.check(lib.lidbox_bn_bwd(nv.ptr(ws.fe_a[i]), dy, R, Co, ctypes.c_void_p(c.data_ptr()), ctypes.c_void_p(c.data_ptr() + 4 * Co),
                                       self._p(l.name + "_bn.gamma"), 1, self._p(l.name + "_bn.gamma", True),
                                       self._p(l.name + "_bn.beta", True), nv.ptr(ws.fe_dz[i]), nv.ptr(ws.bn_ws), ws.bn_ws.numel(), st))
            dz = _rows(ws.fe_dz[i].data_ptr(), 0, Co, 1, R)
            A_rows = self._fe_rows_in(ws, i)
            self._launch_wgrad(ws, lambda w, nb, s_, A_rows=A_rows, dz=dz, l=l, K=K, Co=Co, g=g: nv.check(g.tn(
                A_rows, dz, self._p(l.name + "_conv.W", True), Co, K, Co, 0, self._p(l.name + "_conv.b", True), w, nb, s_)))
            if i == 0:
                continue
            # dgrad into fe_dy[i-1] [B*T, F_i, C_in]: tap groups as in backward_conv_ws, the frame as the "batch"
            dprev = ws.fe_dy[i - 1]
            Fi, Fo, BT = self.fe_dims[i], self.fe_dims[i + 1], ws.B * ws.T
            if l.k < l.s:
                nv.check(lib.lidbox_zero_2d(nv.ptr(dprev), 4 * dprev.numel(), 4 * dprev.numel(), 1, st))
            elif Fo * l.s < Fi:
                nv.check(lib.lidbox_zero_2d(ctypes.c_void_p(dprev.data_ptr() + 4 * Fo * l.s * cin), 4 * Fi * cin,
                                            4 * (Fi - Fo * l.s) * cin, BT, st))
            for grp in range((l.k + l.s - 1) // l.s):
                ntaps = min(l.s, l.k - grp * l.s)
                Cd = _rows(dprev.data_ptr() + 4 * grp * l.s * cin, Fi * cin, l.s * cin, BT, Fo)
                Wg = ctypes.c_void_p(self._p(l.name + "_conv.W").value + 4 * grp * l.s * cin * Co)
                nv.check(g.nt(dz, Wg, Co, Cd, Co, ntaps * cin, nv.EPI_NONE if grp == 0 else nv.EPI_ACCUM, None, gws, gws_n, st))
        self.join_wgrad()

    def fused_output_ok(self):
        """the last Dense + log_softmax + sparse cross-entropy (and their backward) can run as lidbox_softmax_head_fwd_bwd:
        few classes, a plain Dense on the fp32 family behind at least one other Dense"""
        if self.output_activation != "log_softmax" or len(self.denses) < 2 or self.denses[-1].relu:
            return False
        if self.dense_gemm.name != "float32":
            return False
        return bool(nv.lib.lidbox_softmax_head_supported(self.denses[-2].units, self.denses[-1].units))

    def forward_ws(self, ws, upto_embedding=False, training=False, update_moving=True, stop_before_output=False):
        """The model input buffer (ws.input_view()) must already hold the input.  Returns logp (or the embedding).
        training selects batch statistics in BatchNormalization layers (update_moving=False leaves the running
        statistics alone: warm-up passes before a graph capture).  stop_before_output: return the input of the last Dense
        instead (the train step fuses that layer with the loss)."""
        st = nv.current_stream()
        lib = nv.lib
        if self.frontend:
            self._forward_frontend(ws, training, update_moving)
        if self.bf16_storage:
            self._refresh_bf16_weights()
            if ws.act16[0] is not None and not ws.__dict__.pop("input16_fresh", False):   # else: the feature kernel wrote it
                nv.check(lib.lidbox_f32_to_bf16(nv.ptr(ws.act[0]), nv.ptr(ws.act16[0]), ws.act[0].numel(), st))
        cin = self.input_dim
        fresh16 = False                                              # act16[i] holds bf16(act[i]) (written by conv i-1's epilogue)
        for i, c in enumerate(self.convs):
            if ws.B * ws.Ts[i + 1] > 0 and c.d == 1 and ws.act16[i] is not None:
                # bf16-storage path: A = bf16 shadow of act[i], B = transposed bf16 kernel; the epilogue also writes the
                # shadow of act[i+1] when the next conv reads it
                if i > 0 and not fresh16:                            # the producer ran on the fp32-source path
                    nv.check(lib.lidbox_f32_to_bf16(nv.ptr(ws.act[i]), nv.ptr(ws.act16[i]), ws.act[i].numel(), st))
                fresh16 = ws.act16[i + 1] is not None
                out_rows = self._rows_out(ws.act[i + 1], ws, i + 1)
                nxt = ws.act16[i + 1]
                sh = None if nxt is None else ctypes.c_void_p(nxt.data_ptr() + (out_rows.base - ws.act[i + 1].data_ptr()) // 2)
                if self.bf16_only and nxt is not None and i + 1 < len(self.convs):
                    out_rows = nv.Rows(None, out_rows.batch_stride, out_rows.row_stride, out_rows.batch, out_rows.rows_per_batch)
                elif ws.last16 is not None and i + 1 == len(self.convs):
                    # the pooling reads the shadow: no fp32 output, the shadow at its own (8-channel padded) strides
                    l16 = ws.last16
                    out_rows = nv.Rows(None, l16.shape[1] * l16.shape[2], l16.shape[2], out_rows.batch, out_rows.rows_per_batch)
                    sh = nv.ptr(l16)
                nv.check(lib.lidbox_gemm_bf16s_nt(self._rows16(self._conv_rows_in(ws, i), ws.act[i], ws.act16[i]),
                                                  nv.ptr(self.w16t[i]), c.k * cin, out_rows, sh, c.k * cin, c.filters,
                                                  nv.EPI_BIAS_RELU if c.relu else nv.EPI_BIAS, self._p(c.name + ".b"),
                                                  nv.ptr(ws.gemm_ws), ws.gemm_ws.numel(), st))
            elif ws.B * ws.Ts[i + 1] > 0 and c.d == 1:
                fresh16 = False
                nv.check(self.gemm.nn(self._conv_rows_in(ws, i), self._p(c.name + ".W"), c.filters,
                                            self._rows_out(ws.act[i + 1], ws, i + 1), c.k * cin, c.filters,
                                            nv.EPI_BIAS_RELU if c.relu else nv.EPI_BIAS, self._p(c.name + ".b"),
                                            nv.ptr(ws.gemm_ws), ws.gemm_ws.numel(), st))
            elif ws.B * ws.Ts[i + 1] > 0:
                fresh16 = False
                # dilated: tap j reads rows shifted by j*d; the first tap adds the bias, the last applies the ReLU
                out = self._rows_out(ws.act[i + 1], ws, i + 1)
                for j in range(c.k):
                    if j == 0:
                        epi = nv.EPI_BIAS
                    elif j < c.k - 1 or not c.relu:
                        epi = nv.EPI_ACCUM
                    else:
                        epi = nv.EPI_ACCUM_RELU
                    nv.check(self.gemm.nn(self._tap_rows(ws.act[i], ws, i, j), self._tap_weight(c, cin, j), c.filters,
                                          out, cin, c.filters, epi, self._p(c.name + ".b") if j == 0 else None,
                                          nv.ptr(ws.gemm_ws), ws.gemm_ws.numel(), st))
            cin = c.filters
        last = ws.act[-1]
        T, C = last.shape[1], last.shape[2]
        att = self.attention
        if att is not None and ws.B * T > 0:
            n = ws.B * T
            gws, gws_n = nv.ptr(ws.gemm_ws), ws.gemm_ws.numel()
            nv.check(self.gemm.nn(_rows(last.data_ptr(), 0, C, 1, n), self._p("Wf_1.W"), att.d_a,
                                  _rows(ws.fa_x1.data_ptr(), 0, att.d_a, 1, n), C, att.d_a, nv.EPI_RELU, None, gws, gws_n, st))
            nv.check(self.gemm.nn(_rows(ws.fa_x1.data_ptr(), 0, att.d_a, 1, n), self._p("Wf_2.W"), att.d_f,
                                  _rows(ws.fa_F.data_ptr(), 0, att.d_f, 1, n), att.d_a, att.d_f, nv.EPI_NONE, None, gws, gws_n, st))
            nv.check(lib.lidbox_freq_attention_fwd(nv.ptr(last), nv.ptr(ws.fa_F), n, C, att.d_f, nv.ptr(ws.fa_F),
                                                   nv.ptr(ws.hw), st))
            last = ws.hw
        if ws.last16 is not None:
            C16 = ws.last16.shape[2]
            nv.check(lib.lidbox_stats_pool_fwd_bf16(nv.ptr(ws.last16), ws.B, T, C, T * C16, C16, nv.ptr(ws.pooled), st))
        else:
            fn = lib.lidbox_stats_pool_fwd if self.pool == "stats" else lib.lidbox_avg_pool_fwd
            nv.check(fn(nv.ptr(last), ws.B, T, C, T * C, C, nv.ptr(ws.pooled), st))
        x, din = ws.pooled, ws.pooled.shape[1]
        for j, d in enumerate(self.denses):
            if stop_before_output and j == len(self.denses) - 1:
                return x
            emb = upto_embedding and j == 0
            out = ws.emb if emb else ws.h[j]
            epi = nv.EPI_BIAS_RELU if (d.relu and not emb) else nv.EPI_BIAS
            nv.check(self.dense_gemm.nn(_rows(x.data_ptr(), 0, din, 1, ws.B), self._p(d.name + ".W"), d.units,
                                        _rows(out.data_ptr(), 0, d.units, 1, ws.B), din, d.units, epi,
                                        self._p(d.name + ".b"), nv.ptr(ws.gemm_ws), ws.gemm_ws.numel(), st))
            if emb:
                return ws.emb
            x, din = out, d.units
        if self.output_activation is None:
            return ws.h[-1]
        if self.output_activation == "softmax":                      # ws.logp holds the probabilities then
            nv.check(lib.lidbox_softmax_fwd(nv.ptr(ws.h[-1]), ws.B, din, nv.ptr(ws.logp), st))
            return ws.logp
        nv.check(lib.lidbox_log_softmax_fwd(nv.ptr(ws.h[-1]), ws.B, din, nv.ptr(ws.logp), st))
        return ws.logp

    # ------------------------------------------------------------------ backward
    # Carried reduces.  A wgrad leaves M-slices of partial sums; the fixed-order sum that finishes dW / db is bandwidth-bound
    # work that used to sit between two MFMA-bound launches.  The library can run it in the leading workgroups of a later
    # GEMM launch instead (lidbox_gemm_nt_carry / lidbox_gemm_nt_tn_carry; same sums, bit-identical), so the engine keeps
    # the jobs that are still open in ws.pending and hands them to the next dgrad; whatever is left at the end of a
    # backward stage goes out as one launch (flush_reduce_jobs).  LIDBOX_GEMM_NO_CARRY=1 keeps every reduce a launch.
    def _tn_region(self, ws):
        """index of a wgrad workspace no pending job still reads"""
        busy = {r for _, r in ws.pending}
        for r in (0, 1):
            if r not in busy:
                return r
        self.flush_reduce_jobs(ws)
        return 0

    def _take_jobs(self, ws, limit):
        """up to `limit` pending jobs as a ctypes array (oldest first; older extras are flushed), removed from the list"""
        if len(ws.pending) > limit:
            keep = ws.pending[len(ws.pending) - limit:] if limit > 0 else []
            ws.pending = ws.pending[:len(ws.pending) - limit]
            self.flush_reduce_jobs(ws)
            ws.pending = keep
        n = len(ws.pending)
        arr = (nv.ReduceJob * max(1, n))(*[j for j, _ in ws.pending])
        ws.pending = []
        return arr, n

    def flush_reduce_jobs(self, ws):
        """run what is still pending as a launch of its own (end of a backward stage: the gradients must be final)"""
        while ws.pending:
            chunk, ws.pending = ws.pending[:2], ws.pending[2:]
            arr = (nv.ReduceJob * len(chunk))(*[j for j, _ in chunk])
            nv.check(nv.lib.lidbox_reduce_jobs_run(arr, len(chunk), nv.current_stream()))

    def _dgrad_wgrad(self, ws, dy, W, ldb, dX, Co, N, epi, aux, X, dW, ldc, K1, db):
        """a layer's dgrad + wgrad on the fp32 family (lidbox_gemm_nt_tn_carry): carries one pending job of an earlier
        layer, leaves its own reduce pending when the library could not place it inside its launches"""
        r = self._tn_region(ws)
        jobs, n = self._take_jobs(ws, 1)
        out = nv.ReduceJob()
        tws = ws.tn_regions[r]
        nv.check(nv.lib.lidbox_gemm_nt_tn_carry(dy, W, ldb, dX, Co, N, epi, aux, nv.ptr(ws.gemm_ws), ws.gemm_ws.numel(), X, dW, ldc, K1, 0, db,
                                                nv.ptr(tws), tws.numel(), jobs, n, ctypes.byref(out), nv.current_stream()))
        if out.nblocks:
            ws.pending.append((out, r))

    def _launch_wgrad(self, ws, launch, head=False):
        """launch(workspace_ptr, workspace_bytes, stream): on the side stream (after everything enqueued so
        far on the current stream) when one is configured, else inline.  head: a dense-head wgrad -- those few-workgroup
        launches go to `head_wgrad_stream` when only that one is configured (Trainer's default): they then run beside the
        head's dgrad chain instead of between its links."""
        side = self.wgrad_stream
        if side is None and head:
            side = self.head_wgrad_stream
        if side is None:
            launch(nv.ptr(ws.gemm_ws), ws.gemm_ws.numel(), nv.current_stream())
            return
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            launch(nv.ptr(ws.gemm_ws2), ws.gemm_ws2.numel(), nv.current_stream())

    def join_wgrad(self):
        """make the side stream's wgrad results visible to the current stream"""
        if self.wgrad_stream is not None:
            torch.cuda.current_stream().wait_stream(self.wgrad_stream)
        if self.head_wgrad_stream is not None:
            torch.cuda.current_stream().wait_stream(self.head_wgrad_stream)

    def backward_ws(self, ws):
        """dh[-1] must hold d loss / d logits.  Fills flat_grad (overwrites)."""
        self.backward_head_ws(ws)
        for i in range(len(self.convs) - 1, -1, -1):
            self.backward_conv_ws(ws, i)
        self.flush_reduce_jobs(ws)
        self.join_wgrad()

    def backward_head_ws(self, ws):
        """dense chain + pooling backward: fills the dense gradients and dact[-1]."""
        st = nv.current_stream()
        lib = nv.lib
        B = ws.B
        ws.d16_fresh = set()                 # indices j whose dact16[j] holds bf16(dact[j]) (written by a dgrad epilogue)
        ws.pending = []                      # jobs a pass that raised half-way left behind belong to that pass: never carried over
        gws, gws_n = nv.ptr(ws.gemm_ws), ws.gemm_ws.numel()
        # ---- dense chain (the output layer's share is already there when the train step fused it with the loss)
        top = len(self.denses) - 1
        if getattr(ws, "output_layer_done", False):
            ws.output_layer_done = False
            top -= 1
        for j in range(top, -1, -1):
            d = self.denses[j]
            x = ws.pooled if j == 0 else ws.h[j - 1]
            din = x.shape[1]
            dy = _rows(ws.dh[j].data_ptr(), 0, d.units, 1, B)
            A_rows = _rows(x.data_ptr(), 0, din, 1, B)
            dst = ws.dpooled if j == 0 else ws.dh[j - 1]
            relu_prev = j > 0 and self.denses[j - 1].relu
            epi, aux = (nv.EPI_RELU_MASK, nv.ptr(x)) if relu_prev else (nv.EPI_NONE, None)
            dst_rows = _rows(dst.data_ptr(), 0, din, 1, B)
            if self.dense_gemm.name == "float32" and self.wgrad_stream is None and self.head_wgrad_stream is None:
                # wgrad and dgrad of the layer read the same dy: one launch when both are small (lidbox_gemm_nt_tn), the
                # wgrad's partial sums in the second workspace
                self._dgrad_wgrad(ws, dy, self._p(d.name + ".W"), d.units, dst_rows, d.units, din, epi, aux,
                                  A_rows, self._p(d.name + ".W", True), d.units, din, self._p(d.name + ".b", True))
                continue
            self._launch_wgrad(ws, lambda w, n, s_, A_rows=A_rows, dy=dy, d=d, din=din: nv.check(self.dense_gemm.tn(
                A_rows, dy, self._p(d.name + ".W", True), d.units, din, d.units, 0, self._p(d.name + ".b", True), w, n, s_)),
                head=True)
            nv.check(self.dense_gemm.nt(dy, self._p(d.name + ".W"), d.units, dst_rows, d.units, din, epi, aux, gws, gws_n, st))
        # ---- pooling (fused with the ReLU backward of the last conv)
        att = self.attention
        last = ws.act[-1] if att is None else ws.hw
        dlast = ws.dact[-1] if att is None else ws.d_hw
        T, C = last.shape[1], last.shape[2]
        relu_last = 1 if self.convs[-1].relu else 0
        pool_mask = relu_last if att is None else 0          # with attention the pooling input is not a ReLU output
        d16 = ws.dact16[-1] if att is None else None
        if ws.last16 is not None and d16 is not None and B * T > 0:
            C16 = ws.last16.shape[2]
            nv.check(lib.lidbox_stats_pool_bwd_bf16(nv.ptr(ws.last16), nv.ptr(ws.pooled), nv.ptr(ws.dpooled), B, T, C, T * C16, C16,
                                                    pool_mask, nv.ptr(d16), T * d16.shape[2], d16.shape[2], st))
            ws.d16_fresh.add(len(self.convs))
        elif self.pool == "stats" and d16 is not None and B * T > 0:
            # the last conv's output gradient and its bf16 shadow (rows possibly padded) in one pass
            nv.check(lib.lidbox_stats_pool_bwd_shadow(nv.ptr(last), nv.ptr(ws.pooled), nv.ptr(ws.dpooled), B, T, C, T * C, C,
                                                      pool_mask, None if self.bf16_only else nv.ptr(dlast), nv.ptr(d16),
                                                      T * d16.shape[2], d16.shape[2], st))
            ws.d16_fresh.add(len(self.convs))
        elif self.pool == "stats":
            nv.check(lib.lidbox_stats_pool_bwd(nv.ptr(last), nv.ptr(ws.pooled), nv.ptr(ws.dpooled), B, T, C, T * C, C,
                                               pool_mask, nv.ptr(dlast), st))
        else:
            nv.check(lib.lidbox_avg_pool_bwd(nv.ptr(last), nv.ptr(ws.dpooled), B, T, C, T * C, C, pool_mask,
                                             nv.ptr(dlast), st))
        if att is None or B * T == 0:
            return
        # ---- frequency attention (clstm.py:31-42): dact[-1] = mask(H) * (d_hw * F[bin] + d_x1 Wf_1^T)
        n = B * T
        H = ws.act[-1]
        nv.check(lib.lidbox_freq_attention_bwd(nv.ptr(H), nv.ptr(ws.fa_F), nv.ptr(ws.d_hw), n, C, att.d_f, relu_last,
                                               nv.ptr(ws.d_logit), nv.ptr(ws.dact[-1]), st))
        x1r, dlr = _rows(ws.fa_x1.data_ptr(), 0, att.d_a, 1, n), _rows(ws.d_logit.data_ptr(), 0, att.d_f, 1, n)
        Hr, dx1r = _rows(H.data_ptr(), 0, C, 1, n), _rows(ws.d_x1.data_ptr(), 0, att.d_a, 1, n)
        self._launch_wgrad(ws, lambda w, n_, s_: nv.check(self.gemm.tn(
            x1r, dlr, self._p("Wf_2.W", True), att.d_f, att.d_a, att.d_f, 0, None, w, n_, s_)))
        nv.check(self.gemm.nt(dlr, self._p("Wf_2.W"), att.d_f, dx1r, att.d_f, att.d_a, nv.EPI_RELU_MASK,
                              nv.ptr(ws.fa_x1), gws, gws_n, st))
        self._launch_wgrad(ws, lambda w, n_, s_: nv.check(self.gemm.tn(
            Hr, dx1r, self._p("Wf_1.W", True), att.d_a, C, att.d_a, 0, None, w, n_, s_)))
        nv.check(self.gemm.nt(dx1r, self._p("Wf_1.W"), att.d_a, _rows(ws.dact[-1].data_ptr(), 0, C, 1, n), att.d_a, C,
                              nv.EPI_ACCUM_RELU_MASK if relu_last else nv.EPI_ACCUM, nv.ptr(H) if relu_last else None,
                              gws, gws_n, st))

    def backward_conv_ws(self, ws, i):
        """wgrad / bias grad of conv i and (i > 0) dgrad into dact[i]; dact[i+1] must be final."""
        st = nv.current_stream()
        lib = nv.lib
        B = ws.B
        gws, gws_n = nv.ptr(ws.gemm_ws), ws.gemm_ws.numel()
        c = self.convs[i]
        cin = self.input_dim if i == 0 else self.convs[i - 1].filters
        K = c.k * cin
        To = ws.Ts[i + 1]
        if B * To == 0:
            self.param(c.name + ".W", True).zero_()
            self.param(c.name + ".b", True).zero_()
            return
        dy = self._rows_out(ws.dact[i + 1], ws, i + 1)
        if c.d > 1:
            self._backward_dilated(ws, i, dy)
            return
        A_rows = self._conv_rows_in(ws, i)
        pair_wgrad = False
        # bf16 shadow of dact[i+1]: written by conv i+1's dgrad epilogues / the pooling backward, or converted here when it
        # came from a fp32-source launch (same-layout shadows only)
        dy16 = ws.dact16[i + 1] if self.bf16_storage else None
        if dy16 is not None and (i + 1) not in ws.d16_fresh:
            if dy16.shape == ws.dact[i + 1].shape:
                nv.check(lib.lidbox_f32_to_bf16(nv.ptr(ws.dact[i + 1]), nv.ptr(dy16), ws.dact[i + 1].numel(), st))
                ws.d16_fresh.add(i + 1)
            else:
                dy16 = None
        if dy16 is not None and ws.act16[i] is not None and self.shadow_wgrad_ok(i):
            # wgrad on the shadows (transpose-read operands, half the bytes of the fp32-source kernel)
            A16, B16 = self._rows16(A_rows, ws.act[i], ws.act16[i]), self._rows16(dy, ws.dact[i + 1], dy16)
            if self.wgrad_stream is None:
                # the GEMM now; its slice sum rides in this layer's first dgrad launch (or the stage's flush)
                r = self._tn_region(ws)
                job, tws = nv.ReduceJob(), ws.tn_regions[r]
                nv.check(lib.lidbox_gemm_bf16s_tn_partial(A16, B16, self._p(c.name + ".W", True), c.filters, K, c.filters, 0,
                                                          self._p(c.name + ".b", True), nv.ptr(tws), tws.numel(), ctypes.byref(job), st))
                if job.nblocks:
                    ws.pending.append((job, r))
            else:
                self._launch_wgrad(ws, lambda w, n, s_: nv.check(lib.lidbox_gemm_bf16s_tn(
                    A16, B16, self._p(c.name + ".W", True), c.filters, K, c.filters, 0, self._p(c.name + ".b", True), w, n, s_)))
        elif (self.gemm.name == "float32" and self.wgrad_stream is None and not self.bf16_storage and
              not (i == 0 and not self.frontend)):
            pair_wgrad = True             # goes out with the first dgrad group below (lidbox_gemm_nt_tn: one launch when both are small)
        elif self.gemm.name == "float32" and self.wgrad_stream is None and not self.bf16_storage:
            # no dgrad behind it (the first layer): the GEMM now, its reduce with whatever the stage still has pending
            r = self._tn_region(ws)
            job, tws = nv.ReduceJob(), ws.tn_regions[r]
            nv.check(lib.lidbox_gemm_tn_partial(A_rows, dy, self._p(c.name + ".W", True), c.filters, K, c.filters, 0,
                                                self._p(c.name + ".b", True), nv.ptr(tws), tws.numel(), ctypes.byref(job), st))
            if job.nblocks:
                ws.pending.append((job, r))
        else:
            self._launch_wgrad(ws, lambda w, n, s_: nv.check(self.gemm.tn(
                A_rows, dy, self._p(c.name + ".W", True), c.filters, K, c.filters, 0, self._p(c.name + ".b", True), w, n, s_)))
        if i == 0 and not self.frontend:
            return
        # dgrad into dact[i].  Window t touches padded rows [t*s, t*s+k).  Group g = taps
        # [g*s, g*s+ntaps) writes rows (t+g)*s + [0, ntaps): disjoint across t, so each group is one
        # GEMM; group 0 overwrites, later groups accumulate.  Rows no tap produces must read as zero.
        dprev, aprev = ws.dact[i], ws.act[i]
        Tp = dprev.shape[1]
        relu_prev = self.convs[i - 1].relu if i > 0 else False      # conv 0 behind a front-end reads a BatchNorm output
        ngroups = (c.k + c.s - 1) // c.s
        # bf16-storage dgrad: A = bf16 shadow of dact[i+1] (written by conv i+1's dgrad epilogues / the pooling backward, or
        # converted above), B = a bf16 image of the Keras kernel
        use16 = self.bf16_storage and dy16 is not None and self.shadow_dgrad_ok(i)
        d16 = ws.dact16[i] if use16 else None                   # shadow of this dgrad's output, for conv i-1's dgrad
        only16 = self.bf16_only and use16 and d16 is not None   # no fp32 copy of dact[i]; ReLU mask from act16[i]'s signs
        if use16 and ngroups > 1 and self.wd16[i]:
            # k > s, output-stationary (_dgrad_residues): one GEMM per row residue, every row of dact[i] behind the pad written
            # exactly once; windows running past an utterance's last gradient row read the zero trail rows of dact16[i+1]
            Tp2, cp = dy16.shape[1], dy16.shape[2]
            calls = []
            for rho, Q, u_min, u_max in self._dgrad_residues(i, ws.Ts[i]):
                nu, p0 = u_max - u_min + 1, u_min * c.s + rho
                A16 = nv.Rows(dy16.data_ptr() + 2 * (ws.pads[i + 1] + u_min - Q + 1) * cp, Tp2 * cp, cp, B, nu)
                Cd = _rows(None if only16 else dprev.data_ptr() + 4 * p0 * cin, Tp * cin, c.s * cin, B, nu)
                epi, mask = nv.EPI_NONE, None
                if relu_prev and only16:
                    epi, mask = nv.EPI_RELU_MASK | nv.EPI_MASK_BF16, ctypes.c_void_p(ws.act16[i].data_ptr() + 2 * p0 * cin)
                elif relu_prev:
                    epi, mask = nv.EPI_RELU_MASK, ctypes.c_void_p(aprev.data_ptr() + 4 * p0 * cin)
                sh = None if d16 is None else ctypes.c_void_p(d16.data_ptr() + 2 * p0 * cin)
                calls.append((A16, nv.ptr(self.wd16[i][rho]), Q * cp, Cd, sh, Q * cp, cin, epi, mask))
            if len(calls) == 2:
                # the two row residues of a stride-2 layer (frame2: taps {0, 2} and {1}) write disjoint rows: ONE grid when both run
                # on the ping-pong tile (lidbox_gemm_bf16s_nt_pair_carry; otherwise the library issues the two launches itself)
                jobs, nj = self._take_jobs(ws, 2)
                nv.check(lib.lidbox_gemm_bf16s_nt_pair_carry(*calls[0], *calls[1], gws, gws_n, jobs, nj, st))
            else:
                for call in calls:
                    jobs, nj = self._take_jobs(ws, 2)
                    nv.check(lib.lidbox_gemm_bf16s_nt_carry(*call, gws, gws_n, jobs, nj, st))
            if d16 is not None:
                ws.d16_fresh.add(i)
            return
        if use16 and (ngroups > 1 or "fused" not in self.wd16[i]) and c.filters % 8 != 0:
            use16, d16 = False, None                             # grouped taps read flat16 as it is: 8-channel rows only
        # Rows no tap writes read as zero: they are zero since allocation and stay so when a single group writes (k <= s);
        # with several groups the later ones ACCUMULATE into rows group 0 does not cover, which are cleared first.
        if ngroups > 1 and To * c.s < Tp:
            if pair_wgrad and ((Tp - To * c.s) * cin) % 4 == 0 and (Tp * cin) % 4 == 0 and dprev.data_ptr() % 16 == 0 and (To * c.s * cin) % 4 == 0:
                # as a carried job: cleared in the leading workgroups of the group-0 launch below (group 0 does not touch these rows)
                zj = nv.ReduceJob()
                nv.check(lib.lidbox_zero_job(ctypes.c_void_p(dprev.data_ptr() + 4 * To * c.s * cin), Tp * cin, (Tp - To * c.s) * cin, B,
                                             ctypes.byref(zj)))
                self.flush_reduce_jobs(ws)
                ws.pending.append((zj, -1))
            else:
                nv.check(lib.lidbox_zero_2d(ctypes.c_void_p(dprev.data_ptr() + 4 * To * c.s * cin), 4 * Tp * cin,
                                            4 * (Tp - To * c.s) * cin, B, st))
            if d16 is not None:
                nv.check(lib.lidbox_zero_2d(ctypes.c_void_p(d16.data_ptr() + 2 * To * c.s * cin), 2 * Tp * cin,
                                            2 * (Tp - To * c.s) * cin, B, st))
        if d16 is not None:
            ws.d16_fresh.add(i)
        fused16 = self.wd16[i].get("fused") if use16 else None   # [k*C_in][C_out padded to 8]
        for g in range(ngroups):
            ntaps = min(c.s, c.k - g * c.s)
            base_off = 4 * g * c.s * cin                    # bytes
            Cd = _rows(dprev.data_ptr() + base_off, Tp * cin, c.s * cin, B, To)
            Wg = ctypes.c_void_p(self._p(c.name + ".W").value + 4 * g * c.s * cin * c.filters)
            mask = ctypes.c_void_p(aprev.data_ptr() + base_off) if relu_prev else None
            if g == 0:
                epi = nv.EPI_RELU_MASK if relu_prev else nv.EPI_NONE
            else:
                epi = nv.EPI_ACCUM_RELU_MASK if relu_prev else nv.EPI_ACCUM
            if use16:
                sh = None if d16 is None else ctypes.c_void_p(d16.data_ptr() + base_off // 2)
                A16 = self._rows16(dy, ws.dact[i + 1], dy16)
                if only16 and ngroups == 1:
                    Cd = _rows(None, Tp * cin, c.s * cin, B, To)
                    if relu_prev:
                        epi, mask = epi | nv.EPI_MASK_BF16, ctypes.c_void_p(ws.act16[i].data_ptr() + base_off // 2)
                jobs, nj = self._take_jobs(ws, 2)
                if fused16 is not None:                          # padded channel rows on both operands (pad columns are zero)
                    cp = fused16.shape[1]
                    nv.check(lib.lidbox_gemm_bf16s_nt_carry(A16, nv.ptr(fused16), cp, Cd, sh, cp, ntaps * cin, epi, mask, gws, gws_n, jobs, nj, st))
                else:
                    Wg16 = ctypes.c_void_p(self._p16(c.name + ".W").value + 2 * g * c.s * cin * c.filters)
                    nv.check(lib.lidbox_gemm_bf16s_nt_carry(A16, Wg16, c.filters, Cd, sh, c.filters, ntaps * cin, epi, mask, gws, gws_n,
                                                            jobs, nj, st))
            elif pair_wgrad and g == 0:
                self._dgrad_wgrad(ws, dy, Wg, c.filters, Cd, c.filters, ntaps * cin, epi, mask,
                                  A_rows, self._p(c.name + ".W", True), c.filters, K, self._p(c.name + ".b", True))
            else:
                nv.check(self.gemm.nt(dy, Wg, c.filters, Cd, c.filters, ntaps * cin, epi, mask, gws, gws_n, st))
        if i == 0:
            self._backward_frontend(ws)

    def _backward_dilated(self, ws, i, dy):
        """dilated conv i (strides 1): per tap j, dW[j] = act[i][rows + j*d]^T dY and
        dact[i][rows + j*d] += mask * dY W[j]^T; the bias gradient comes with tap 0."""
        st = nv.current_stream()
        gws, gws_n = nv.ptr(ws.gemm_ws), ws.gemm_ws.numel()
        c = self.convs[i]
        cin = self.input_dim if i == 0 else self.convs[i - 1].filters
        for j in range(c.k):
            A_rows = self._tap_rows(ws.act[i], ws, i, j)
            self._launch_wgrad(ws, lambda w, n, s_, A_rows=A_rows, j=j: nv.check(self.gemm.tn(
                A_rows, dy, self._tap_weight(c, cin, j, True), c.filters, cin, c.filters, 0,
                self._p(c.name + ".b", True) if j == 0 else None, w, n, s_)))
        if i == 0:
            return
        dprev, aprev = ws.dact[i], ws.act[i]
        relu_prev = self.convs[i - 1].relu
        To = ws.Ts[i + 1]
        if To < dprev.shape[1]:
            Tp = dprev.shape[1]                             # rows tap 0 does not overwrite
            nv.check(nv.lib.lidbox_zero_2d(ctypes.c_void_p(dprev.data_ptr() + 4 * To * cin), 4 * Tp * cin,
                                           4 * (Tp - To) * cin, ws.B, st))
        for j in range(c.k):
            mask = ctypes.c_void_p(aprev.data_ptr() + 4 * j * c.d * cin) if relu_prev else None
            if j == 0:
                epi = nv.EPI_RELU_MASK if relu_prev else nv.EPI_NONE
            else:
                epi = nv.EPI_ACCUM_RELU_MASK if relu_prev else nv.EPI_ACCUM
            nv.check(self.gemm.nt(dy, self._tap_weight(c, cin, j), c.filters, self._tap_rows(dprev, ws, i, j), c.filters,
                                  cin, epi, mask, gws, gws_n, st))

    # ------------------------------------------------------------------ public call
    def _load_input(self, ws, x, training):
        x = nv.require_gpu_tensor(x, "x", torch.float32)
        if x.dim() != 3 or x.shape[2] != self.model_input_dim:
            raise ValueError("expected input [B, T, %d], got %s" % (self.model_input_dim, tuple(x.shape)))
        if x.stride(2) != 1 or x.stride(1) != x.shape[2]:
            x = x.contiguous()
        st = nv.current_stream()
        in_ptr, in_bs, _, C = ws.input_target()
        nv.check(nv.lib.lidbox_copy_2d(in_ptr, 4 * in_bs, nv.ptr(x), 4 * (x.stride(0) if ws.B > 1 else ws.T * C),
                                       4 * ws.T * C, ws.B, st))
        if training and self.channel_dropout_rate > 0:
            # Keras SpatialDropout1D (xvector.py:50-51): whole channels dropped per utterance; eager calls draw from a
            # host-side call counter (the captured train step keys its masks on the device-side Adam step instead)
            self._dropout_calls += 1
            nv.check(nv.lib.lidbox_spatial_dropout(in_ptr, ws.B, ws.T, C, in_bs, self.channel_dropout_rate,
                                                   (self.dropout_seed + 0x51ED27 * self._dropout_calls) & (2 ** 64 - 1),
                                                   None, None, st))

    def __call__(self, x, training=False):
        """x [B,T,C] on the HIP device -> log-probs [B, num_outputs] (a fresh tensor)."""
        with torch.cuda.device(self.device):
            ws = self.workspace(x.shape[0], x.shape[1])
            self._load_input(ws, x, training)
            return self.forward_ws(ws, training=training).clone()

    def embed(self, x):
        """as_embedding_extractor output: first dense layer's affine output (activation removed)."""
        with torch.cuda.device(self.device):
            ws = self.workspace(x.shape[0], x.shape[1])
            self._load_input(ws, x, False)
            return self.forward_ws(ws, upto_embedding=True).clone()

    predict = __call__


class EmbeddingExtractor:
    """Result of `as_embedding_extractor(model)` (reference xvector.py:70-73 / cnn.py:19-22)."""

    def __init__(self, model):
        self.model = model
        self.name = model.name + ":" + model.embedding_layer

    def __call__(self, x, training=False):
        return self.model.embed(x)
