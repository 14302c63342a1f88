"""
Counterpart of lidbox/models/xvector.py: x-vector TDNN -- five causal strided Conv1D+ReLU frame
layers, mean+stddev pooling over time, two Dense+ReLU segment layers, Dense, log_softmax
(reference xvector.py:46-67).  NOTE: the reference passes its third
`frame_layer` argument to Keras as STRIDES, not dilation, and uses no BatchNorm; this build
follows the reference.
"""
from .tdnn import ConvSpec, DenseSpec, EmbeddingExtractor, SequentialTDNN

TIME_AXIS = 1
STDDEV_SQRT_MIN_CLIP = 1e-10     # reference xvector.py:22 (applied inside lidbox_stats_pool_fwd)


class GlobalMeanStddevPooling1D:
    """reference xvector.py:25-35: mean and standard deviation of the inputs [B, T, C] over the time axis, concatenated
    to [B, 2C]; the variance is clipped to [1e-10, max] before the square root.  A callable over device tensors on
    `lidbox_stats_pool_fwd` (the kernel the models' pooling stage runs); time-major rows may be strided."""
    name = "stats_pooling"

    def __call__(self, inputs):
        import torch
        from .. import _native as nv
        x = nv.require_gpu_tensor(inputs, "inputs", torch.float32)
        if x.dim() != 3:
            raise ValueError("expected inputs [B, T, C], got %s" % (tuple(x.shape),))
        if x.stride(2) != 1:
            x = x.contiguous()
        B, T, C = x.shape
        if T == 0:
            raise ValueError("cannot pool over an empty time axis")
        out = torch.empty((B, 2 * C), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            nv.check(nv.lib.lidbox_stats_pool_fwd(nv.ptr(x), B, T, C, x.stride(0) if B > 1 else T * x.stride(1), x.stride(1),
                                                  nv.ptr(out), nv.current_stream()))
        return out

    call = __call__


def frame_layer(filters, kernel_size, strides, padding="causal", activation="relu", name="frame", dilation_rate=1):
    """reference xvector.py:38-39.  dilation_rate is not in the reference (its third argument is Keras `strides`);
    it is this build's opt-in for Kaldi-style dilated TDNN contexts and follows Keras Conv1D(dilation_rate=...).
    padding: "causal" (what every reference model uses), "valid" or "same", as Keras Conv1D takes them."""
    if activation not in ("relu", None):
        raise ValueError("activation must be 'relu' or None")
    return ConvSpec(name, filters, kernel_size, strides, relu=(activation == "relu"), dilation_rate=dilation_rate, padding=padding)


def segment_layer(units, activation="relu", name="segment"):
    """reference xvector.py:42-43"""
    if activation not in ("relu", None):
        raise ValueError("activation must be 'relu' or None")
    return DenseSpec(name, units, relu=(activation == "relu"))


def create(input_shape, num_outputs, channel_dropout_rate=0, name="x-vector", seed=None, device=None,
           compute_dtype="float32"):
    """reference xvector.py:46-67.  input_shape = (T or None, C); returns a callable model:
    model(x [B,T,C], training=bool) -> log-probs [B, num_outputs].
    compute_dtype (not in the reference; Keras sets this through a global mixed-precision policy): "float32" or
    "bfloat16" = GEMM operands rounded to bf16 on chip, fp32 accumulate, fp32 master weights (BASELINE config 5)."""
    convs = [
        frame_layer(512, 5, 1, name="frame1"),
        frame_layer(512, 3, 2, name="frame2"),
        frame_layer(512, 3, 3, name="frame3"),
        frame_layer(512, 1, 1, name="frame4"),
        frame_layer(1500, 1, 1, name="frame5"),
    ]
    denses = [
        segment_layer(512, name="segment1"),
        segment_layer(512, name="segment2"),
        DenseSpec("outputs", num_outputs, relu=False),
    ]
    return SequentialTDNN(input_shape, convs, "stats", denses, name=name, output_activation="log_softmax",
                          channel_dropout_rate=channel_dropout_rate, seed=seed, device=device, compute_dtype=compute_dtype)


loader = create      # lidbox/models/keras_utils.py:134 calls `model_module.loader(...)`


def as_embedding_extractor(m):
    """reference xvector.py:70-73: output of `segment1` with its activation removed, [B, 512]."""
    return EmbeddingExtractor(m)
