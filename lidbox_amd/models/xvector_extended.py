"""
Counterpart of lidbox/models/xvector_extended.py (reference :22-43): the extended x-vector of
Villalba et al. (2018) -- ten causal strided Conv1D+ReLU frame layers (strides 1/2/3/4), mean+stddev
pooling, two Dense+ReLU segment layers, Dense, log_softmax.  Same kernels as `xvector`; the
k=3 / stride-4 layer (frame7) exercises the dgrad path where some input rows feed no tap.
"""
from .tdnn import DenseSpec, SequentialTDNN
from .xvector import as_embedding_extractor, frame_layer, segment_layer  # noqa: F401

FRAMES = [(512, 5, 1), (512, 1, 1), (512, 3, 2), (512, 1, 1), (512, 3, 3), (512, 1, 1), (512, 3, 4), (512, 1, 1),
          (512, 1, 1), (1500, 1, 1)]          # (filters, kernel_size, strides) of frame1..frame10, reference :25-34


def create(input_shape, num_outputs, output_activation="log_softmax", seed=None, device=None, compute_dtype="float32"):
    """reference xvector_extended.py:22-43"""
    convs = [frame_layer(f, k, s, name="frame%d" % (i + 1)) for i, (f, k, s) in enumerate(FRAMES)]
    denses = [segment_layer(512, name="segment1"), segment_layer(512, name="segment2"),
              DenseSpec("output", num_outputs, relu=False)]
    return SequentialTDNN(input_shape, convs, "stats", denses, name="x-vector-extended",
                          output_activation=output_activation, seed=seed, device=device, compute_dtype=compute_dtype)


loader = create
