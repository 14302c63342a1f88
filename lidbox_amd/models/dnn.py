"""
Counterpart of lidbox/models/dnn.py (reference :13-22): four time-distributed Dense+ReLU layers (200, 400, 600, 800
units; a Dense on [B, T, C] is a pointwise Conv1D and runs on the same implicit-row GEMMs), GlobalAveragePooling1D,
Dense, log_softmax.
"""
from .tdnn import ConvSpec, DenseSpec, SequentialTDNN


def create(input_shape, num_outputs, seed=None, device=None, compute_dtype="float32"):
    """reference dnn.py:13-22"""
    convs = [ConvSpec("fc_%d" % (i + 1), units, 1, 1, relu=True, dense_kernel=True) for i, units in enumerate((200, 400, 600, 800))]
    return SequentialTDNN(input_shape, convs, "avg", [DenseSpec("output", num_outputs, relu=False)], name="DNN",
                          output_activation="log_softmax", seed=seed, device=device, compute_dtype=compute_dtype)


loader = create
