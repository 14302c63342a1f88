"""
Train step for the hot path: counterpart of what `KerasWrapper.from_config` + `fit` make Keras do
per batch (reference lidbox/models/keras_utils.py:124-149, 191-203): forward(training=True) ->
loss (mean over the batch) -> gradients -> Adam(lr 1e-3, betas 0.9/0.999, epsilon 1e-7) ->
optional streaming metrics.  Here the step optionally starts from raw waveforms (the fused
log-mel / MFCC kernel writes straight into the first Conv1D's input buffer) and the whole step
is captured into hipGraphs.

Data parallelism (new in this build; the reference is single-device): one process per GPU,
utterances sharded contiguously across ranks, weights replicated, ONE exchange per step: an
all-reduce(sum) of the flat gradient buffer over RCCL (torch.distributed backend "nccl"),
then Adam scales by 1/world.  The gradient buffer is reduced in contiguous buckets; each is
issued as an asynchronous collective as soon as its last wgrad has been enqueued and overlaps
the remaining backward GEMMs.
"""
import ctypes
import gc
import os

import torch

from . import _native as nv
from .losses import SparseAngularProximity


# ------------------------------------------------------------------ distributed helpers
def init_distributed(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun).
    Returns (rank, world_size, local_rank).  No-op for a single process."""
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or os.environ.get("LIDBOX_FORCE_GRAD_SYNC")) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
    return rank, world, local_rank


def shard_bounds(global_batch, rank, world):
    """Contiguous utterance shard [lo, hi) of rank `rank`; remainders go to the first ranks."""
    q, r = divmod(int(global_batch), int(world))
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


class GradSync:
    """Bucketed all-reduce(sum) of one flat gradient tensor.

    `bucket_bounds` are element offsets [0 = b0 < b1 < ... < bn = numel]; bucket i is
    flat[b_i:b_{i+1}].  Buckets are reduced in the order they are `launch`ed (backward fills the
    HIGH offsets first).  Each launch is an asynchronous collective: torch.distributed orders it after
    everything enqueued so far on the current HIP stream (an event on that stream, waited for by the
    backend's own communication stream) and runs it concurrently with whatever the compute stream is given
    next, so it overlaps the rest of backward; `wait()` makes the compute stream wait for all of them.
    (An explicit side stream + event per bucket did the same with two more cross-stream hops per bucket and was
    10 us per step slower.)  On CPU (gloo, used by the CPU tests) the same calls apply.
    """

    def __init__(self, flat, bucket_bounds, group=None, wire_dtype=None):
        import torch.distributed as dist
        self.dist = dist
        self.flat, self.bounds, self.group = flat, list(bucket_bounds), group
        assert self.bounds[0] == 0 and self.bounds[-1] == flat.numel()
        assert all(a < b for a, b in zip(self.bounds, self.bounds[1:]))
        # wire format of the exchange.  None / float32: the gradient buffer itself is all-reduced in place (18.2 MB per step for
        # the 100-language x-vector).  "bfloat16": a bucket is rounded ONCE to a bf16 staging buffer (lidbox_f32_to_bf16), the
        # staging buffer is all-reduced (half the bytes on the xGMI links: what the 1.3 ms bf16-compute step of configs[4]
        # asks for), and widened back into the fp32 gradient buffer for Adam (lidbox_bf16_to_f32).  The sum over ranks is then
        # formed in bf16 by the collective: ~3 significant digits per element, the precision of the bf16 backward that
        # produced the gradients; fp32 master weights and Adam moments are untouched.
        wire_formats = {None: None, "float32": None, "bfloat16": torch.bfloat16, torch.float32: None, torch.bfloat16: torch.bfloat16}
        if wire_dtype not in wire_formats:
            raise ValueError("grad_wire_dtype must be None, 'float32' or 'bfloat16', got %r" % (wire_dtype,))
        wire_dtype = wire_formats[wire_dtype]
        if wire_dtype is not None and flat.dtype != torch.float32:
            raise ValueError("a bfloat16 wire format needs a float32 gradient buffer")
        self.wire_dtype = wire_dtype
        self.wire = torch.zeros(flat.numel(), dtype=wire_dtype, device=flat.device) if wire_dtype is not None else None
        self.active = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        if os.environ.get("LIDBOX_FORCE_GRAD_SYNC") and dist.is_available() and dist.is_initialized():
            self.active = True           # test aid: exercise the collective path even at world_size 1
        self.world = dist.get_world_size(group) if self.active else 1
        self.cuda = flat.is_cuda
        self._pending = []
        self.wait_events = None          # a list: `wait` appends (before, after) HIP events of the compute stream (bench.py)
        # RCCL collectives can be captured into a hipGraph (torch.distributed's nccl backend records them on its own
        # communication stream, forked from / joined to the capturing stream); gloo stages through the host and cannot
        self.capturable = bool(self.active and self.cuda and dist.get_backend(group) == "nccl")

    def warm(self):
        """one eager collective on the gradient buffer's device: the communicator must exist before a capture starts"""
        if self.active:
            t = torch.zeros(1, dtype=self.flat.dtype, device=self.flat.device)
            self.dist.all_reduce(t, group=self.group)

    @property
    def num_buckets(self):
        return len(self.bounds) - 1

    def launch(self, i):
        """all-reduce bucket i; everything enqueued on the current stream so far is visible to it."""
        if not self.active:
            return
        lo, hi = self.bounds[i], self.bounds[i + 1]
        view = self.flat[lo:hi]
        if self.wire is not None:
            w = self.wire[lo:hi]
            self._narrow(view, w)
            self._pending.append((self.dist.all_reduce(w, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True), lo, hi))
            return
        self._pending.append((self.dist.all_reduce(view, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True), lo, hi))

    def _narrow(self, src32, dst16):
        """fp32 -> wire format on the current stream (round-to-nearest-even); bucket bounds keep both 16-byte aligned"""
        if self.cuda:
            nv.check(nv.lib.lidbox_f32_to_bf16(nv.ptr(src32), nv.ptr(dst16), src32.numel(), nv.current_stream()))
        else:
            dst16.copy_(src32)                   # gloo on host tensors (CPU tests): torch's conversion rounds the same way

    def _widen(self, src16, dst32):
        if self.cuda:
            nv.check(nv.lib.lidbox_bf16_to_f32(nv.ptr(src16), nv.ptr(dst32), src16.numel(), nv.current_stream()))
        else:
            dst32.copy_(src16)

    def wait(self):
        """make the reduced gradients visible to the compute stream (HIP) / the caller (CPU)"""
        ev = None
        if self.wait_events is not None and self.cuda and self._pending:
            # measurement aid (bench.py, eager steps only): HIP events on the compute stream around the joins.  The stream has
            # nothing else to do between them, so their distance is the part of the exchange the backward did NOT hide
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        for work, lo, hi in self._pending:
            work.wait()
            if self.wire is not None:
                self._widen(self.wire[lo:hi], self.flat[lo:hi])
        if ev is not None:
            ev[1].record()
            self.wait_events.append(ev)
        self._pending.clear()

    @property
    def wire_bytes(self):
        """bytes one rank hands to the collectives per step"""
        return self.flat.numel() * (2 if self.wire is not None else self.flat.element_size())

    @property
    def grad_scale(self):
        """1 / world: what turns the all-reduced SUM of per-rank mean gradients into their mean (equal shards only).
        `Trainer` does not use it: it scales every rank's loss gradient by 1 / global_batch instead (`_loss_scale`),
        which is also right for uneven shards, and hands Adam a grad_scale of 1."""
        return 1.0 / self.world


def plan_buckets(model, num_buckets=2):
    """Split the flat gradient into `num_buckets` contiguous buckets at conv-layer boundaries.  Backward fills the
    flat buffer from its HIGH end, so the highest bucket is complete first and is reduced while the lower layers'
    backward GEMMs still run; the lowest bucket is the only one whose reduction is exposed, so it is kept small.
    The first split is the conv boundary closest to a quarter of the parameters (for the x-vector: frame1 + frame2
    below, 3.6 MB); further buckets peel single conv layers off the low end (three buckets: frame1 | frame2 | rest,
    exposed all-reduce 0.4 MB).  Returns (bounds, splits): bounds = element offsets [0, ..., numel]; splits =
    ascending conv indices, bucket j+1 is complete once backward_conv_ws(splits[j]) has been enqueued; for two
    buckets `splits` is the single index itself (None when there is one bucket)."""
    if num_buckets < 2 or len(model.convs) < 2:
        return [0, model.num_flat], None
    total = model.num_flat
    best, best_i = None, None
    for i in range(1, len(model.convs)):
        off = model.layout[model.convs[i].name + ".W"][0]
        score = abs(off - total * 0.25)          # lower layers finish last: keep the tail bucket small
        if best is None or score < best:
            best, best_i = score, i
    splits = [best_i]
    i = best_i - 1
    while len(splits) < num_buckets - 1 and i >= 1:
        splits.insert(0, i)
        i -= 1
    bounds = [0] + [model.layout[model.convs[i].name + ".W"][0] for i in splits] + [total]
    return bounds, (splits[0] if num_buckets == 2 else splits)


# ------------------------------------------------------------------ trainer
class Trainer:
    """One compiled train step.

    model      : lidbox_amd.models.tdnn.SequentialTDNN (xvector.create / cnn.create)
    loss       : "sparse_categorical_crossentropy" (Keras SparseCategoricalCrossentropy(from_logits=True)
                 on the log-softmax outputs, keras_utils.py:141-142) or a SparseAngularProximity
                 instance (applied to the L2-normalised model output; the model must then end in an
                 affine layer with output_activation=None)
    optimizer  : dict(lr, beta_1, beta_2, epsilon) -- tf.keras.optimizers.Adam defaults
    feature    : None (inputs are features [B,T,C]) or dict(plan=FeaturePlan, kind=nv.FEAT_*)
                 (inputs are waveforms [B,N]; features are computed inside the step)
    """

    def __init__(self, model, loss="sparse_categorical_crossentropy", optimizer=None, feature=None,
                 use_graph=True, num_buckets=2, group=None, metric=None, overlap_wgrad=False, overlap_head_wgrad=False,
                 sync_state_every_step=False, grad_wire_dtype=None):
        self.model = model
        self.device = model.device
        # optimizer: dict(cls="Adam" | "SGD" | "RMSprop", ...) with the Keras argument names and defaults (TensorFlow 2.3); the class
        # is what a config names at keras_utils.py:137-140
        cls_ = (optimizer or {}).get("cls", "Adam")
        defaults = {"Adam": dict(lr=1e-3, beta_1=0.9, beta_2=0.999, epsilon=1e-7),
                    "SGD": dict(lr=0.01, momentum=0.0, nesterov=False),
                    "RMSprop": dict(lr=1e-3, rho=0.9, momentum=0.0, epsilon=1e-7, centered=False)}
        if cls_ not in defaults:
            raise ValueError("unsupported optimizer %r (Adam, SGD, RMSprop)" % (cls_,))
        opt = dict(defaults[cls_], cls=cls_)
        opt.update(optimizer or {})
        unknown = set(opt) - set(defaults[cls_]) - {"cls", "lr_schedule"}
        if unknown:
            raise ValueError("%s: unknown optimizer arguments %s" % (cls_, sorted(unknown)))
        self.opt = opt
        self.feature = feature
        self.metric = metric
        if isinstance(loss, SparseAngularProximity):
            if model.output_activation is not None:
                raise ValueError("angular proximity loss needs a model without output activation")
            self.loss_kind, self.ap = "ap", loss
        elif loss == "sparse_categorical_crossentropy":
            if model.output_activation != "log_softmax":
                raise ValueError("sparse_categorical_crossentropy expects log_softmax outputs")
            self.loss_kind, self.ap = "nll", None
        elif loss == "sparse_categorical_crossentropy_probs":
            # Keras SparseCategoricalCrossentropy(from_logits=False): the model must end in tf.nn.softmax
            if model.output_activation != "softmax":
                raise ValueError("SparseCategoricalCrossentropy(from_logits=False) expects a model built with output_activation='softmax'")
            self.loss_kind, self.ap = "nll_probs", None
        else:
            raise ValueError("unknown loss %r" % (loss,))
        f32 = dict(dtype=torch.float32, device=self.device)
        self.m = torch.zeros(model.num_flat, **f32)
        self.v = torch.zeros(model.num_flat, **f32)
        if opt["cls"] == "RMSprop" and opt["centered"]:
            self.mg = torch.zeros(model.num_flat, **f32)         # centered RMSprop's running mean of the gradient (allocated here: never inside a capture)
        self.adam_state = torch.zeros(16, dtype=torch.uint8, device=self.device)     # {int64 step, float lr_t, float lr_now}
        bounds, splits = plan_buckets(model, num_buckets)
        self.splits = [] if splits is None else ([splits] if isinstance(splits, int) else list(splits))   # ascending conv indices
        self.split_conv = self.splits[-1] if self.splits else None
        # grad_wire_dtype: "bfloat16" halves the bytes of the gradient exchange (GradSync); default: the fp32 buffer in place
        self.sync = GradSync(model.flat_grad, bounds, group, wire_dtype=grad_wire_dtype)
        # overlap_wgrad: run wgrad GEMMs on a second stream concurrently with the dgrad chain.  Measured neutral
        # (96.1k vs 97.1k utt/s at bs 256): both are bound by the same matrix pipes.  Off by default.
        # fuse_output: the last Dense, log_softmax, the cross-entropy and their backward as two small launches when the model allows
        # it (SequentialTDNN.fused_output_ok); LIDBOX_NO_FUSED_OUTPUT=1 keeps the nine separate launches (A/B aid)
        import os as _os
        self.fuse_output = self.loss_kind == "nll" and _os.environ.get("LIDBOX_NO_FUSED_OUTPUT") != "1" and model.fused_output_ok()
        if _os.environ.get("LIDBOX_OVERLAP_WGRAD") is not None:          # A/B aid
            overlap_wgrad = _os.environ["LIDBOX_OVERLAP_WGRAD"] == "1"
        if overlap_wgrad and model.wgrad_stream is None:
            model.wgrad_stream = torch.cuda.Stream(device=self.device)
        # overlap_head_wgrad: only the dense head's wgrads (a few workgroups each, M = batch) on a second stream, beside the
        # head's dgrad chain.  Measured SLOWER inside the captured step (fp32 2.387 -> 2.439 ms, bf16 0.913 -> 0.951 ms at
        # bs 256): the three fork / join edges cost the graph more than the ~35 us of small kernels they could hide.  Off.
        if overlap_head_wgrad and model.head_wgrad_stream is None:
            model.head_wgrad_stream = torch.cuda.Stream(device=self.device)
        self.sync_state_every_step = bool(sync_state_every_step)
        self.use_graph = bool(use_graph)
        self._warming = False            # True during the pre-capture warm-up pass: streaming metrics must not count it
        self._graphs = {}
        self._static = {}
        self._first_batch = None         # (local shard size, global batch) of the first step without an explicit global_batch
        self._step_global_batch = None   # what the caller of the running step passed (train_step(..., global_batch=))
        self.grad_sync_mode = "none"     # what the last built step does with the gradient exchange: none | in_graph | segmented | eager
        self._size_stream = None
        self._feat_ready = {}            # id(workspace) -> (data_ptr, shape of the waveforms whose features a prefetch left, buffer parity)
        self._prefetch_stream = None
        self._step_feat = None           # (skip_inline_features, next_inputs or None, parity of the prefetch target) of the step being issued

    @property
    def adam_state(self):
        return self._adam_state

    @adam_state.setter
    def adam_state(self, t):
        """{int64 step, float lr_t, float lr_now} on the device.  Re-pointing it (bench.py shares one state between a graph
        and an eager trainer) invalidates the host's view of it: the learning-rate schedule re-reads the device step."""
        self._adam_state = t
        self.__dict__.pop("_lr_view", None)
        self.__dict__.pop("_host_step", None)

    # ---------------------------------------------------------------- pieces of a step
    def _forward_loss(self, ws, inputs, labels):
        lib, st = nv.lib, nv.current_stream()
        model = self.model
        # FrameLayer2D dropout masks (tdnn._dropout_step_ptr / _fe_dropout_seed) follow THIS trainer's optimizer step and
        # rank: bound when the step is issued / captured, not when a Trainer is constructed (a second Trainer on the same
        # model -- bench.py's eager one -- must not re-key the first one's masks)
        model.dropout_step = self.adam_state
        model.dropout_seed_mix = self._rank_mix()
        skip_inline, nxt, nxt_parity = self._step_feat or (False, None, 0)
        if nxt is not None:
            # feature prefetch: the NEXT batch's features, extracted on a second stream beside this step's MFMA-bound GEMMs into
            # the other input buffer (vector-ALU-bound work under matrix-pipe-bound work; joined behind the forward pass)
            if self._prefetch_stream is None:
                self._prefetch_stream = torch.cuda.Stream(device=self.device)
            cur_parity = 1 - nxt_parity
            ws.select_input_buffer(nxt_parity)
            side = self._prefetch_stream
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self._extract_features(ws, nxt, nv.current_stream())
            ws.select_input_buffer(cur_parity)
        elif hasattr(ws, "_in_bufs"):
            ws.select_input_buffer(0)
        in_ptr, in_bs, _, C = ws.input_target()              # act[0] behind the first Conv1D's causal zero rows, or the 2-D front-end's input
        if self.feature is not None:
            if not skip_inline:
                self._extract_features(ws, inputs, st)
        else:
            if inputs.stride(2) != 1 or inputs.stride(1) != C:
                inputs = inputs.contiguous()
            nv.check(lib.lidbox_copy_2d(in_ptr, 4 * in_bs, nv.ptr(inputs), 4 * (inputs.stride(0) if ws.B > 1 else ws.T * C),
                                        4 * ws.T * C, ws.B, st))
        if model.channel_dropout_rate > 0:
            # SpatialDropout1D of the training pass (xvector.py:50-51, cnn.py:29-30); the Adam step counter on the device
            # keys the mask, so every replay of the captured step draws a new one
            nv.check(lib.lidbox_spatial_dropout(in_ptr, ws.B, ws.T, C, in_bs, model.channel_dropout_rate,
                                                self._dropout_seed(), nv.ptr(self.adam_state), None, st))
        ws.input16_fresh = self._input_shadow_from_features(ws)
        # BatchNormalization layers: batch statistics; the running statistics move once per real step (not in warm-up passes)
        B = ws.B
        scale = self._loss_scale(B)
        if self.loss_kind == "nll" and self.fuse_output and B > 0:
            # output Dense + log_softmax + cross-entropy and their backward in two small launches (few classes: latency, not work)
            x = model.forward_ws(ws, training=True, update_moving=not self._warming, stop_before_output=True)
            d = model.denses[-1]
            K, N = x.shape[1], d.units
            if not hasattr(ws, "head_ws"):
                ws.head_ws = torch.empty(nv.lib.lidbox_softmax_head_workspace(B, K, N), dtype=torch.uint8, device=self.device)
            nv.check(lib.lidbox_softmax_head_fwd_bwd(nv.ptr(x), model._p(d.name + ".W"), model._p(d.name + ".b"), nv.ptr(labels), B, K, N,
                                                     scale, 1 if model.denses[-2].relu else 0, nv.ptr(ws.logp), nv.ptr(ws.loss),
                                                     model._p(d.name + ".W", True), model._p(d.name + ".b", True), nv.ptr(ws.dh[-2]),
                                                     nv.ptr(ws.head_ws), ws.head_ws.numel(), st))
            ws.output_layer_done = True
            out = ws.logp
        else:
            out = model.forward_ws(ws, training=True, update_moving=not self._warming)
        if self.loss_kind == "nll":
            if not getattr(ws, "output_layer_done", False):
                nv.check(lib.lidbox_nll_fwd_bwd(nv.ptr(out), nv.ptr(labels), B, out.shape[1], scale,
                                                nv.ptr(ws.loss), nv.ptr(ws.dh[-1]), st))
        elif self.loss_kind == "nll_probs":
            # clipped-probability cross-entropy and its gradient with respect to the logits, from the logits (ws.h[-1])
            nv.check(lib.lidbox_softmax_nll_fwd_bwd(nv.ptr(ws.h[-1]), nv.ptr(labels), B, out.shape[1], scale, None,
                                                    nv.ptr(ws.loss), nv.ptr(ws.dh[-1]), st))
        else:
            D = out.shape[1]
            zn, dzn, per = self._ap_buffers(ws, D)
            want_scores = self.metric is not None and not self._warming
            if D <= 4096 and not os.environ.get("LIDBOX_AP_SEPARATE"):
                # normalise -> loss + gradient -> gradient through the normalisation -> predict() scores: one launch
                nv.check(lib.lidbox_ap_head_fwd_bwd(nv.ptr(out), nv.ptr(labels), B, D, self.ap.N, self.ap.delta_weight, scale, None,
                                                    nv.ptr(per), nv.ptr(ws.dh[-1]), nv.ptr(ws.ap_scores) if want_scores else None, st))
            else:
                nv.check(lib.lidbox_l2_normalize_fwd(nv.ptr(out), B, D, nv.ptr(zn), st))
                nv.check(lib.lidbox_ap_loss_fwd_bwd(nv.ptr(zn), nv.ptr(labels), B, D, self.ap.N, self.ap.delta_weight,
                                                    scale, nv.ptr(per), nv.ptr(dzn), st))
                nv.check(lib.lidbox_l2_normalize_bwd(nv.ptr(out), nv.ptr(dzn), B, D, nv.ptr(ws.dh[-1]), st))
                if want_scores:
                    nv.check(lib.lidbox_neg_acos(nv.ptr(zn), B, D, self.ap.N, nv.ptr(ws.ap_scores), st))
            nv.check(lib.lidbox_mean(nv.ptr(per), B, nv.ptr(ws.loss), st))
            if want_scores:
                self.metric._update_sparse(labels, ws.ap_scores)
        if self.loss_kind in ("nll", "nll_probs") and self.metric is not None and not self._warming:
            self.metric._update_sparse(labels, out)
        if nxt is not None:
            torch.cuda.current_stream().wait_stream(self._prefetch_stream)      # the prefetch joins behind the forward pass

    def _extract_features(self, ws, signals, st):
        """waveforms -> the model's input buffer (ws.act[0] as currently selected): fused log-mel / MFCC kernel, CMVN in place"""
        lib = nv.lib
        in_ptr, in_bs, _, C = ws.input_target()
        plan, kind = self.feature["plan"], self.feature["kind"]
        Bn, N = signals.shape
        stride = signals.stride(0) if Bn > 1 else N
        # features land directly in the model's input buffer -- and, on the bf16-storage path, in its bf16 shadow as well
        sh = None
        if self._input_shadow_from_features(ws):
            sh = ctypes.c_void_p(ws.act16[0].data_ptr() + (in_ptr.value - ws.act[0].data_ptr()) // 2)
        nv.check(lib.lidbox_extract_features_fwd_shadow(plan.handle, kind, nv.ptr(signals), Bn, N, stride, in_ptr, in_bs, sh, None, 0, st))
        if self.feature.get("cmvn"):
            # per-utterance CMVN over time (features/__init__.py:22-32), in place where the conv reads it
            nv.check(lib.lidbox_cmvn_strided_fwd(in_ptr, Bn, ws.T, C, in_bs, 1, in_ptr, in_bs, st))

    def _input_shadow_from_features(self, ws):
        """True when the feature call also writes bf16(act[0]) (lidbox_extract_features_fwd_shadow: the log-mel kernel's store
        stage does it), so that forward_ws skips its conversion launch: bf16-storage model, 1-D input, and nothing rewriting
        the features in place afterwards (CMVN, channel dropout)."""
        m = self.model
        if self.feature is None or not m.bf16_storage or m.frontend or ws.act16[0] is None:
            return False
        if self.feature.get("cmvn") or m.channel_dropout_rate > 0 or os.environ.get("LIDBOX_FEAT_NO_SHADOW"):    # env: A/B aid
            return False
        in_ptr, _, _, _ = ws.input_target()
        return (in_ptr.value - ws.act[0].data_ptr()) % 16 == 0 and ws.act16[0].data_ptr() % 8 == 0

    def _rank_mix(self):
        """per-rank offset of every dropout seed under data parallelism (0 for a single process)"""
        if not self.sync.active:
            return 0
        return (0x9E3779B97F4A7C15 * (self.sync.dist.get_rank(self.sync.group) + 1)) & (2 ** 63 - 1)

    def _dropout_seed(self):
        """the model's dropout seed, offset per data-parallel rank: the mask of local row b must not repeat on every rank"""
        return (self.model.dropout_seed + self._rank_mix()) & (2 ** 63 - 1)

    def _loss_scale(self, B):
        """d(mean loss)/d(per-example loss).  Under data parallelism the mean is over the GLOBAL batch, so that uneven
        shards (`shard_bounds` gives the remainder to the first ranks) still yield the gradient of the global-batch mean
        after the all-reduce(sum).  The global batch is what the caller passed to `train_step(..., global_batch=)`.
        Without it the ranks' shard sizes are summed ONCE, at every rank's first step (a collective every rank enters:
        rank-symmetric), and that sum stands for as long as this rank's shard size stays what it was then.  A shard size
        that changes later needs the explicit argument: whether the other ranks' sizes changed too cannot be known without
        a collective, and a collective only SOME ranks enter would pair with the others' gradient all-reduce (a hang or a
        mismatched exchange) -- so that case raises instead.  `global_batch_of` is the rank-symmetric exchange callers use
        per step (`KerasWrapper.fit` does).  The scale is a kernel argument, so it is part of the key of a captured step."""
        if not self.sync.active:
            return 1.0 / B
        if self._step_global_batch is not None:
            return 1.0 / self._step_global_batch
        if self._first_batch is None:
            self._first_batch = (B, self.global_batch_of(B))
        if B != self._first_batch[0]:
            raise ValueError("data parallelism: this rank's shard size changed from %d to %d and no global_batch was passed; "
                             "pass train_step(..., global_batch=) (Trainer.global_batch_of(B) computes it on every rank)"
                             % (self._first_batch[0], B))
        return 1.0 / self._first_batch[1]

    def global_batch_of(self, B):
        """sum of the ranks' shard sizes for the coming step: ONE tiny all-reduce that EVERY rank must enter (call it on
        every rank or on none).  It runs on its own stream so that the host waits for the exchange only, not for the
        train steps still queued on the compute stream."""
        if not self.sync.active:
            return int(B)
        if self.sync.cuda:
            if self._size_stream is None:
                self._size_stream = torch.cuda.Stream(device=self.device)
            with torch.cuda.stream(self._size_stream):
                t = torch.tensor([int(B)], dtype=torch.int64, device=self.device)
                self.sync.dist.all_reduce(t, group=self.sync.group)
                return int(t.item())
        t = torch.tensor([int(B)], dtype=torch.int64)
        self.sync.dist.all_reduce(t, group=self.sync.group)
        return int(t.item())

    def _ap_buffers(self, ws, D):
        if not hasattr(ws, "ap_zn"):
            ws.ap_zn = torch.zeros((ws.B, D), dtype=torch.float32, device=self.device)
            ws.ap_dzn = torch.zeros_like(ws.ap_zn)
            ws.ap_per = torch.zeros(ws.B, dtype=torch.float32, device=self.device)
            ws.ap_scores = torch.zeros((ws.B, self.ap.N), dtype=torch.float32, device=self.device)
        return ws.ap_zn, ws.ap_dzn, ws.ap_per

    def _backward_stage(self, ws, k):
        """Stage k of backward (k = 0 first).  Stage 0: dense head + pooling + the conv layers down to the highest
        split; stage j: the conv layers between two splits; the last stage ends at conv 0.  After stage k the
        gradient bucket (num_buckets - 1 - k) is complete on the current stream."""
        convs = self.model.convs
        hi_edges = [len(convs)] + self.splits[::-1]              # exclusive upper conv index of every stage
        lo_edges = self.splits[::-1] + [0]
        if k == 0:
            self.model.backward_head_ws(ws)
        for i in range(hi_edges[k] - 1, lo_edges[k] - 1, -1):
            self.model.backward_conv_ws(ws, i)
        self._prepared = False
        if k == self.num_stages - 1 and not self._warming and self.opt["cls"] == "Adam" and not os.environ.get("LIDBOX_ADAM_PREPARE_LAUNCH"):
            # the optimizer's scalar half (step counter, bias-corrected rate) rides in the launch that finishes the last wgrad
            job = nv.ReduceJob()
            o = self.opt
            nv.check(nv.lib.lidbox_adam_prepare_job(nv.ptr(self.adam_state), o["lr"], o["beta_1"], o["beta_2"], ctypes.byref(job)))
            ws.pending.append((job, -1))
            self._prepared = True
        self.model.flush_reduce_jobs(ws)       # carried wgrad reduces still open: the stage's gradient bucket must be final
        self.model.join_wgrad()

    @property
    def num_stages(self):
        return len(self.splits) + 1

    def _backward_hi(self, ws):
        """two-bucket view kept for callers / tests: everything down to the highest split"""
        self._backward_stage(ws, 0)

    def _backward_lo(self, ws):
        for k in range(1, self.num_stages):
            self._backward_stage(ws, k)

    def _set_lr(self):
        """learning-rate schedule (opt["lr_schedule"]: optimizer step, 0-based as Keras' `iterations` -> lr): the value of
        the coming step goes into the device-side Adam state, where the captured optimizer kernels read it.  The step
        index is the device's own counter, read once (and again whenever `adam_state` is re-pointed) and then advanced
        on the host only after a step has been issued (`_lr_advance`), so a step that raises does not shift the schedule.
        A scheduled rate of exactly 0 is written as -0.0: an all-zero word means "no schedule" to the kernel."""
        sched = self.opt.get("lr_schedule")
        if sched is None:
            return
        if "_lr_view" not in self.__dict__:
            self._lr_view = self.adam_state[12:16].view(torch.float32)
            self._host_step = self.step_count
        lr = float(sched(self._host_step))
        self._lr_view.fill_(lr if lr != 0.0 else -0.0)

    def _lr_advance(self):
        if "_host_step" in self.__dict__:
            self._host_step += 1

    def _adam(self):
        """the optimizer segment of a step (Adam: the name of the reference configurations' optimizer stuck to it)"""
        o = self.opt
        m = self.model
        if o["cls"] == "SGD":
            nv.check(nv.lib.lidbox_sgd_step(nv.ptr(m.flat), nv.ptr(m.flat_grad), nv.ptr(self.m) if o["momentum"] > 0 else None, m.num_flat,
                                            o["lr"], o["momentum"], int(bool(o["nesterov"])), 1.0, nv.ptr(self.adam_state), nv.current_stream()))
            return
        if o["cls"] == "RMSprop":
            nv.check(nv.lib.lidbox_rmsprop_step(nv.ptr(m.flat), nv.ptr(m.flat_grad), nv.ptr(self.v), nv.ptr(self.mg) if o["centered"] else None,
                                                nv.ptr(self.m) if o["momentum"] > 0 else None, m.num_flat, o["lr"], o["rho"], o["momentum"],
                                                o["epsilon"], int(bool(o["centered"])), 1.0, nv.ptr(self.adam_state), nv.current_stream()))
            return
        if getattr(self, "_prepared", False):      # the last backward stage carried the prepare half (lidbox_adam_prepare_job)
            nv.check(nv.lib.lidbox_adam_apply(nv.ptr(m.flat), nv.ptr(m.flat_grad), nv.ptr(self.m), nv.ptr(self.v), m.num_flat,
                                              o["beta_1"], o["beta_2"], o["epsilon"], 1.0, nv.ptr(self.adam_state), nv.current_stream()))
            return
        nv.check(nv.lib.lidbox_adam_step(nv.ptr(m.flat), nv.ptr(m.flat_grad), nv.ptr(self.m), nv.ptr(self.v),
                                         m.num_flat, o["lr"], o["beta_1"], o["beta_2"], o["epsilon"],
                                         1.0, nv.ptr(self.adam_state), nv.current_stream()))

    # ---------------------------------------------------------------- graph plumbing
    def _capture(self, fn, with_collectives=False):
        # The cyclic collector must not run while the stream is capturing: freeing a dead Trainer's graphs / streams
        # from inside a capture is an illegal HIP call in global capture mode and aborts the process (seen when a
        # test's garbage was collected during the next test's capture).  torch.cuda.graph() collects once on entry.
        g = torch.cuda.CUDAGraph()
        gc_was_enabled = gc.isenabled()
        gc.disable()
        try:
            # a capture that contains collectives runs beside the backend's watchdog thread, whose event queries a
            # "global" capture would treat as errors of THIS capture: thread-local error mode there
            mode = dict(capture_error_mode="thread_local") if with_collectives else {}
            with torch.cuda.graph(g, pool=self._pool(), **mode):
                fn()
        finally:
            if gc_was_enabled:
                gc.enable()
        return g

    def _pool(self):
        if not hasattr(self, "_mempool"):
            self._mempool = torch.cuda.graph_pool_handle()
        return self._mempool

    def _build(self, inputs, labels):
        """warm up eagerly (also validates arguments), then capture the step's graph segments"""
        B = inputs.shape[0]
        T = self.feature["plan"].num_frames(inputs.shape[1]) if self.feature is not None else inputs.shape[1]
        ws = self.model.workspace(B, T)
        # segment 0 = forward + loss + backward stage 0; segment k = backward stage k; last segment = Adam.  With
        # gradient sync active a bucket's all-reduce is launched between two segments.
        segs = [lambda: (self._forward_loss(ws, inputs, labels), self._backward_stage(ws, 0))]
        for k in range(1, self.num_stages):
            segs.append(lambda k=k: self._backward_stage(ws, k))
        segs.append(self._adam)
        entry = dict(ws=ws, inputs=inputs, labels=labels, eager=tuple(segs), graphs=None)
        if self.use_graph:
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                # warm-up on a side stream (required before capture); keeps optimizer and metric state untouched
                self._warming = True
                try:
                    for seg in segs[:-1]:
                        seg()
                finally:
                    self._warming = False
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize(self.device)
            if self.sync.active and self.sync.capturable and not os.environ.get("LIDBOX_SEGMENTED_SYNC"):
                # RCCL: the bucket all-reduces are captured INSIDE the step's graph (fork to the backend's communication
                # stream behind the stage that completes the bucket, join ahead of Adam): an N-GPU step is one graph
                # launch like the single-GPU step.  LIDBOX_SEGMENTED_SYNC=1 keeps the host-launched collectives between
                # graph segments (the only form gloo supports).
                def whole():
                    nb = self.sync.num_buckets
                    for k in range(self.num_stages):
                        segs[k]()
                        if k < nb:
                            self.sync.launch(nb - 1 - k)
                    self.sync.wait()
                    segs[-1]()
                self.sync.warm()
                torch.cuda.synchronize(self.device)
                try:
                    entry["graphs"] = (self._capture(whole, with_collectives=True).replay,)
                    self.grad_sync_mode = "in_graph"
                except RuntimeError as exc:   # a backend build that cannot capture its collectives (HIP / RCCL errors surface as RuntimeError)
                    if os.environ.get("LIDBOX_REQUIRE_INGRAPH_SYNC"):
                        raise RuntimeError("LIDBOX_REQUIRE_INGRAPH_SYNC is set and capturing the gradient all-reduces into the "
                                           "step graph failed: %s" % (exc,)) from exc
                    import warnings
                    warnings.warn("capturing the gradient all-reduces into the step graph failed (%s: %s); falling back to "
                                  "host-launched collectives between graph segments" % (type(exc).__name__, exc))
                    self.sync._pending.clear()
                    torch.cuda.synchronize(self.device)
                    adam_seg = self._capture(segs[-1]).replay if os.environ.get("LIDBOX_ADAM_GRAPH") else segs[-1]
                    entry["graphs"] = tuple(self._capture(seg).replay for seg in segs[:-1]) + (adam_seg,)
                    self.grad_sync_mode = "segmented"
            elif self.sync.active:
                # the optimizer segment is two small kernels behind the last collective: launched directly unless
                # LIDBOX_ADAM_GRAPH is set (a graph launch costs more than it saves there)
                if os.environ.get("LIDBOX_REQUIRE_INGRAPH_SYNC"):
                    raise RuntimeError("LIDBOX_REQUIRE_INGRAPH_SYNC is set but the gradient exchange of this step cannot be captured "
                                       "(backend %r, LIDBOX_SEGMENTED_SYNC=%r)" % (self.sync.dist.get_backend(self.sync.group),
                                                                                  os.environ.get("LIDBOX_SEGMENTED_SYNC")))
                adam_seg = self._capture(segs[-1]).replay if os.environ.get("LIDBOX_ADAM_GRAPH") else segs[-1]
                entry["graphs"] = tuple(self._capture(seg).replay for seg in segs[:-1]) + (adam_seg,)
                self.grad_sync_mode = "segmented"
            else:
                entry["graphs"] = (self._capture(lambda: [seg() for seg in segs]).replay,)
                self.grad_sync_mode = "none"
        else:
            self.grad_sync_mode = "eager" if self.sync.active else "none"
        # whether the CAPTURED backward carries the optimizer's prepare half: the eager Adam segment of the segmented form must
        # follow what its own graphs do, not what the most recent trace (possibly another entry's warm-up) left in _prepared
        entry["prepared"] = bool(getattr(self, "_prepared", False)) if entry["graphs"] is not None else None
        return entry

    # ---------------------------------------------------------------- public API
    def train_step(self, inputs, labels, global_batch=None, next_inputs=None):
        """One optimisation step on this rank's shard.  inputs: waveforms [B,N] (feature != None) or
        features [B,T,C]; labels int32 [B].  Both must stay alive and at the same address between
        calls that reuse the captured graph (pass the same tensors, refilled in place).
        global_batch (data parallelism): the number of utterances of this step over ALL ranks; pass it whenever the ranks'
        shard sizes do not all change together (see `_loss_scale`).
        next_inputs (waveform input only): the waveforms of the NEXT step.  Their features are then extracted during this
        step, on a second stream beside its GEMMs, into the other half of a double-buffered model input, and the next
        call -- which must pass exactly that tensor as `inputs` -- skips its own extraction (the counterpart of the tf.data
        prefetch around the reference's feature step, lidbox/data/steps.py:725-736).  Every batch's features are still
        computed exactly once, one step early; a call whose `inputs` were not prefetched extracts them in line.
        Returns the device scalar holding this rank's mean loss."""
        inputs = nv.require_gpu_tensor(inputs, "inputs", torch.float32)
        labels = nv.require_gpu_tensor(labels, "labels", torch.int32)
        if labels.dim() != 1 or labels.shape[0] != inputs.shape[0]:
            raise ValueError("labels must be [batch_size]")
        key = (inputs.data_ptr(), labels.data_ptr(), tuple(inputs.shape), tuple(inputs.stride()),
               None if global_batch is None else int(global_batch))
        self._step_global_batch = None if global_batch is None else int(global_batch)
        feat_mode, ws_key = None, None
        if self.feature is not None and not self.model.frontend:
            ws_key = (int(inputs.shape[0]), int(self.feature["plan"].num_frames(inputs.shape[1])))
            if next_inputs is not None:
                next_inputs = nv.require_gpu_tensor(next_inputs, "next_inputs", torch.float32)
                if tuple(next_inputs.shape) != tuple(inputs.shape) or tuple(next_inputs.stride()) != tuple(inputs.stride()):
                    raise ValueError("next_inputs must have the shape and strides of inputs")
                ready = self._feat_ready.get(ws_key)
                have = ready is not None and ready[0] == inputs.data_ptr()
                parity = ready[1] if have else 0                  # the buffer this step's features are (or will be) in
                feat_mode = (have, next_inputs, 1 - parity)
                key = key + (next_inputs.data_ptr(), have, parity)
        elif next_inputs is not None:
            raise ValueError("next_inputs needs a Trainer with a waveform feature front-end (and no 2-D front-end)")
        self._step_feat = feat_mode
        with torch.cuda.device(self.device):
            entry = self._graphs.get(key)
            if entry is None:
                if len(self._graphs) >= 16:
                    self._graphs.pop(next(iter(self._graphs)))
                entry = self._build(inputs, labels)
                self._graphs[key] = entry
            ws = entry["ws"]
            self._set_lr()
            if entry["prepared"] is not None:
                self._prepared = entry["prepared"]
            if entry["graphs"] is not None and len(entry["graphs"]) == 1:
                entry["graphs"][0]()
            else:
                run = entry["eager"] if entry["graphs"] is None else entry["graphs"]      # callables: graph replays
                nb = self.sync.num_buckets
                for k in range(self.num_stages):
                    run[k]()
                    if k < nb:
                        self.sync.launch(nb - 1 - k)              # bucket (nb-1-k) is complete after stage k
                self.sync.wait()
                run[-1]()                                         # Adam
            self._lr_advance()
            if ws_key is not None:
                if feat_mode is not None:
                    self._feat_ready[ws_key] = (next_inputs.data_ptr(), feat_mode[2])
                else:
                    self._feat_ready.pop(ws_key, None)
            self._step_feat = None
            if self.sync_state_every_step:
                self.sync_state()
            return ws.loss[0]

    def sync_state(self):
        """BatchNormalization running statistics under data parallelism: every replica normalises with ITS shard's batch
        statistics (tf.keras BatchNormalization under MirroredStrategy; the reference never asks for
        SyncBatchNormalization) and moves its running statistics with them; the replicas' running statistics are
        averaged (the MEAN aggregation Keras declares for these variables) WHEN THEY ARE CONSUMED -- call this before
        evaluating or checkpointing (`KerasWrapper.fit` does at the end of every epoch) -- so every rank then holds the
        same values.  `Trainer(sync_state_every_step=True)` restores the per-step all-reduce."""
        state = getattr(self.model, "state", None)
        if not self.sync.active or not getattr(self.model, "state_layout", None):
            return
        self.sync.dist.all_reduce(state, op=self.sync.dist.ReduceOp.SUM, group=self.sync.group)
        nv.check(nv.lib.lidbox_scale(nv.ptr(state), state.numel(), 1.0 / self.sync.world, nv.current_stream()))

    def loss_and_grads(self, inputs, labels):
        """forward + backward only (no all-reduce, no optimizer): returns (loss, flat_grad view)."""
        inputs = nv.require_gpu_tensor(inputs, "inputs", torch.float32)
        labels = nv.require_gpu_tensor(labels, "labels", torch.int32)
        with torch.cuda.device(self.device):
            B = inputs.shape[0]
            T = self.feature["plan"].num_frames(inputs.shape[1]) if self.feature is not None else inputs.shape[1]
            ws = self.model.workspace(B, T)
            self._feat_ready.pop((int(B), int(T)), None)          # the probe may overwrite a prefetched input buffer
            self._step_feat = None
            was = self._warming
            self._warming = True                 # a probe, not a step: BatchNormalization running statistics and streaming metrics stay put
            try:
                self._forward_loss(ws, inputs, labels)
            finally:
                self._warming = was
            self.model.backward_ws(ws)
            return ws.loss[0], self.model.flat_grad

    @property
    def step_count(self):
        return int(self.adam_state[:8].view(torch.int64).item())
