"""
Counterpart of lidbox/models/xvector_freq_attention.py (reference :22-43): the x-vector with
`clstm.frequency_attention` (reference clstm.py:31-42) between frame5 and the stats pooling -- two bias-free
Dense layers (Wf_1: 1500 -> 64 + ReLU, Wf_2: 64 -> d_f + softmax) produce one weight per frequency bin and
frame; the 1500 channels are scaled in d_f consecutive bins.  The Dense layers run on the GEMM family, the
softmax / bin scaling and their backward are `lidbox_freq_attention_fwd/_bwd` (csrc/attention.hip).
"""
from .tdnn import DenseSpec, FreqAttentionSpec, SequentialTDNN
from .xvector import as_embedding_extractor, frame_layer, segment_layer  # noqa: F401


def create(input_shape, num_outputs, output_activation="log_softmax", freq_attention_bins=60, seed=None, device=None,
           compute_dtype="float32"):
    """reference xvector_freq_attention.py:22-43 (frequency_attention is called with its default d_a=64)"""
    convs = [
        frame_layer(512, 5, 1, name="frame1"),
        frame_layer(512, 3, 2, name="frame2"),
        frame_layer(512, 3, 3, name="frame3"),
        frame_layer(512, 1, 1, name="frame4"),
        frame_layer(1500, 1, 1, name="frame5"),
    ]
    denses = [segment_layer(512, name="segment1"), segment_layer(512, name="segment2"),
              DenseSpec("output", num_outputs, relu=False)]
    return SequentialTDNN(input_shape, convs, "stats", denses, name="x-vector-frequency-attention",
                          output_activation=output_activation, seed=seed, device=device, compute_dtype=compute_dtype,
                          attention=FreqAttentionSpec(d_a=64, d_f=freq_attention_bins))


loader = create
