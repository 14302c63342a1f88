"""
Writes the Keras-layout HDF5 fixtures of tests/test_hdf5_reader.py with h5py (the library Keras itself writes weight
files with).  Run with an interpreter that has h5py, e.g. in this image:

    /opt/conda/bin/python3.9 tests/golden/make_keras_h5.py

(TensorFlow / Keras are not available here, so the files restate the layout of
keras.saving.hdf5_format.save_weights_to_hdf5_group / save_model_to_hdf5: root or "model_weights" attributes
`layer_names`, `backend`, `keras_version`; one group per layer with a `weight_names` attribute; one contiguous float32
dataset per variable at <layer>/<variable name>, e.g. "frame1/frame1/kernel:0".)  Values are an exact integer hash of
(name, index), so the test regenerates them without h5py or a random generator.
"""
import json
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def values(name, shape):
    """deterministic float32 values in [-0.5, 0.5): exact integer arithmetic, no RNG / libm involved"""
    n = int(np.prod(shape))
    seed = zlib.crc32(name.encode("utf-8"))
    i = np.arange(n, dtype=np.uint64)
    h = (i * np.uint64(2654435761) + np.uint64(seed)) % np.uint64(1 << 32)
    v = (h.astype(np.float64) / float(1 << 32) - 0.5).astype(np.float32).reshape(shape)
    if name.endswith("moving_variance:0") or name.endswith("gamma:0"):
        v = v + np.float32(1.0)
    return v


# (layer name, [(variable name as Keras reports it, shape)])
TDNN_LAYERS = [
    ("input", []),
    ("frame1", [("frame1/kernel:0", (5, 6, 8)), ("frame1/bias:0", (8,))]),
    ("frame2", [("frame2/kernel:0", (3, 8, 8)), ("frame2/bias:0", (8,))]),
    ("stats_pooling", []),
    ("segment1", [("segment1/kernel:0", (16, 10)), ("segment1/bias:0", (10,))]),
    ("outputs", [("outputs/kernel:0", (10, 3)), ("outputs/bias:0", (3,))]),
    ("log_softmax", []),
]

FRONTEND_LAYERS = [
    ("input", []),
    ("reshape_to_image", []),
    ("frame2d_1", [("frame2d_1/frame2d_1_conv/kernel:0", (1, 3, 1, 4)), ("frame2d_1/frame2d_1_conv/bias:0", (4,)),
                   ("frame2d_1/frame2d_1_bn/gamma:0", (4,)), ("frame2d_1/frame2d_1_bn/beta:0", (4,)),
                   ("frame2d_1/frame2d_1_bn/moving_mean:0", (4,)), ("frame2d_1/frame2d_1_bn/moving_variance:0", (4,))]),
    ("frame2d_2", [("frame2d_2/frame2d_2_conv/kernel:0", (1, 3, 4, 2)), ("frame2d_2/frame2d_2_conv/bias:0", (2,)),
                   ("frame2d_2/frame2d_2_bn/gamma:0", (2,)), ("frame2d_2/frame2d_2_bn/beta:0", (2,)),
                   ("frame2d_2/frame2d_2_bn/moving_mean:0", (2,)), ("frame2d_2/frame2d_2_bn/moving_variance:0", (2,))]),
    ("flatten_channels", []),
    ("frame1", [("frame1/kernel:0", (3, 6, 8)), ("frame1/bias:0", (8,))]),
    ("stats_pooling", []),
    ("segment1", [("segment1/kernel:0", (16, 5)), ("segment1/bias:0", (5,))]),
    ("output", [("output/kernel:0", (5, 3)), ("output/bias:0", (3,))]),
    ("log_softmax", []),
]


def save_weights_to_group(f, layers, fixed_length_strings=True):
    """fixed_length_strings: numpy "S" arrays / scalars, what Keras + h5py 2.x store for lists of bytes; otherwise the
    variable-length strings h5py 3.x stores for the same Python objects"""
    def names(xs):
        xs = [x.encode("utf8") for x in xs]
        return np.array(xs, dtype="S") if fixed_length_strings and xs else xs
    f.attrs["layer_names"] = names([name for name, _ in layers])
    f.attrs["backend"] = np.bytes_(b"tensorflow") if fixed_length_strings else "tensorflow"
    f.attrs["keras_version"] = np.bytes_(b"2.4.0") if fixed_length_strings else "2.4.0"
    for name, weights in sorted(layers, key=lambda x: x[0]):
        g = f.create_group(name)
        g.attrs["weight_names"] = names([w for w, _ in weights])
        for wname, shape in weights:
            val = values(wname, shape)
            d = g.create_dataset(wname, val.shape, dtype=val.dtype)
            d[:] = val


def main():
    import h5py
    # 1. model.save_weights("...h5"): layer groups under the root
    with h5py.File(os.path.join(HERE, "keras_tdnn_weights.h5"), "w") as f:
        save_weights_to_group(f, TDNN_LAYERS)
    # 1b. the same written with libver="latest" (superblock 3, version 2 object headers, link messages)
    with h5py.File(os.path.join(HERE, "keras_tdnn_weights_latest.h5"), "w", libver="latest") as f:
        save_weights_to_group(f, TDNN_LAYERS)
    # 2. model.save("...hdf5") / ModelCheckpoint: model_weights + optimizer_weights + JSON configs
    with h5py.File(os.path.join(HERE, "keras_frontend_checkpoint.hdf5"), "w") as f:
        f.attrs["keras_version"] = "2.4.0"                                             # variable-length strings (h5py >= 3 + str)
        f.attrs["backend"] = "tensorflow"
        f.attrs["model_config"] = json.dumps({"class_name": "Functional", "config": {"name": "x-vector-2D", "layers": [
            {"name": n} for n, _ in FRONTEND_LAYERS]}})
        f.attrs["training_config"] = json.dumps({"loss": "sparse_categorical_crossentropy"}).encode("utf8")
        save_weights_to_group(f.create_group("model_weights"), FRONTEND_LAYERS, fixed_length_strings=False)
        og = f.create_group("optimizer_weights")
        og.attrs["weight_names"] = [b"Adam/iter:0", b"Adam/frame1/kernel/m:0"]
        og.create_dataset("Adam/iter:0", data=np.int64(17))
        og.create_dataset("Adam/frame1/kernel/m:0", data=values("Adam/frame1/kernel/m:0", (3, 6, 8)))
    print("h5py", h5py.__version__, "HDF5", h5py.version.hdf5_version, "->", sorted(p for p in os.listdir(HERE) if "keras_" in p))


if __name__ == "__main__":
    sys.exit(main())
