"""
Counterpart of lidbox/models/cnn.py (MGB-3 CNN, reference cnn.py:25-45): four Conv1D+ReLU (padding "causal" by
default, "valid" / "same" as Keras Conv1D takes them), global average pooling over time, Dense 1500, Dense 600,
Dense num_outputs, log_softmax.  Same GEMM kernels as the x-vector; only the pooling differs.
"""
from .tdnn import ConvSpec, DenseSpec, EmbeddingExtractor, SequentialTDNN


def create(input_shape, num_outputs, output_activation="log_softmax", padding="causal", channel_dropout_rate=0,
           seed=None, device=None, compute_dtype="float32"):
    """reference cnn.py:25-45"""
    convs = [
        ConvSpec("conv_1", 500, 5, 1, padding=padding),
        ConvSpec("conv_2", 500, 7, 2, padding=padding),
        ConvSpec("conv_3", 500, 1, 1, padding=padding),
        ConvSpec("conv_4", 3000, 1, 1, padding=padding),
    ]
    denses = [DenseSpec("fc_1", 1500), DenseSpec("fc_2", 600), DenseSpec("output", num_outputs, relu=False)]
    return SequentialTDNN(input_shape, convs, "avg", denses, name="MGB-3_CNN", output_activation=output_activation,
                          channel_dropout_rate=channel_dropout_rate, seed=seed, device=device, compute_dtype=compute_dtype)


loader = create


def as_embedding_extractor(keras_model):
    """reference cnn.py:19-22: `fc_1` output with its activation removed."""
    return EmbeddingExtractor(keras_model)
