"""
Counterpart of lidbox/models/xvector_2d.py (reference :66-93): the x-vector with a 2-D CNN front-end -- the input
[B, T, F] becomes a one-channel image, four FrameLayer2D blocks (Conv2D along FREQUENCY with (1, k) kernels and (1, s)
strides, ReLU inside the convolution, then BatchNormalization; reference :26-46) shrink the F axis to `cols` positions of
32 channels, those are flattened to cols * 32 features per frame and fed to the five causal Conv1D frame layers, the
mean + stddev pooling and the Dense head of `xvector`.

Every frame is an independent row of the frequency convolutions, so they run as the same implicit-row MFMA GEMMs as the
Conv1D layers ("batch" = B * T, no padding); BatchNormalization runs on `csrc/batchnorm.hip`.  Under data parallelism the
batch statistics are per replica (what tf.keras BatchNormalization does under MirroredStrategy; SyncBatchNormalization is
not what the reference instantiates) -- see DESIGN.md for the moving-statistics policy.
"""
from .tdnn import DenseSpec, FreqConvSpec, SequentialTDNN
from .xvector import GlobalMeanStddevPooling1D, as_embedding_extractor, frame_layer, segment_layer  # noqa: F401


class FrameLayer2D(FreqConvSpec):
    """reference xvector_2d.py:26-46.  kernel_size / strides are the reference's (1, k) / (1, s) pairs (or plain ints for
    the frequency axis); only padding="valid", activation="relu" -- what the reference's `create` uses."""

    def __init__(self, filters, kernel_size, strides, name="frame", activation="relu", padding="valid", dropout_rate=None):
        k = kernel_size[1] if isinstance(kernel_size, (tuple, list)) else kernel_size
        s = strides[1] if isinstance(strides, (tuple, list)) else strides
        if isinstance(kernel_size, (tuple, list)) and kernel_size[0] != 1 or isinstance(strides, (tuple, list)) and strides[0] != 1:
            raise ValueError("FrameLayer2D convolves along frequency only: kernel (1, k), strides (1, s)")
        if padding != "valid" or activation != "relu":
            raise ValueError("only padding='valid', activation='relu' are supported")
        super().__init__(name, filters, k, s, dropout_rate=dropout_rate)


def create(input_shape, num_outputs, output_activation="log_softmax", seed=None, device=None, compute_dtype="float32"):
    """reference xvector_2d.py:66-93; returns a callable model: model(x [B, T, F], training=bool) -> [B, num_outputs]."""
    frontend = [
        FrameLayer2D(256, (1, 5), (1, 1), name="frame2d_1"),
        FrameLayer2D(128, (1, 3), (1, 2), name="frame2d_2"),
        FrameLayer2D(64, (1, 3), (1, 3), name="frame2d_3"),
        FrameLayer2D(32, (1, 3), (1, 3), name="frame2d_4"),
    ]
    convs = [
        frame_layer(512, 5, 1, name="frame1"),
        frame_layer(512, 3, 2, name="frame2"),
        frame_layer(512, 3, 3, name="frame3"),
        frame_layer(512, 1, 1, name="frame4"),
        frame_layer(1500, 1, 1, name="frame5"),
    ]
    denses = [segment_layer(512, name="segment1"), segment_layer(512, name="segment2"),
              DenseSpec("output", num_outputs, relu=False)]
    return SequentialTDNN(input_shape, convs, "stats", denses, name="x-vector-2D", output_activation=output_activation,
                          seed=seed, device=device, compute_dtype=compute_dtype, frontend=frontend)


loader = create
