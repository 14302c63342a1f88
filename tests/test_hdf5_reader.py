"""
The Keras HDF5 importer (reference lidbox/models/keras_utils.py:57-64, 186-188) against fixtures written by
h5py 3.3 / HDF5 1.10.6 (tests/golden/make_keras_h5.py; the script and the three files are committed).  CPU only.
"""
import os
import sys
import zlib

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLDEN)


def _values(name, shape):
    """tests/golden/make_keras_h5.py: values() restated (exact integer hash, no RNG)"""
    n = int(np.prod(shape))
    i = np.arange(n, dtype=np.uint64)
    h = (i * np.uint64(2654435761) + np.uint64(zlib.crc32(name.encode("utf-8")))) % np.uint64(1 << 32)
    v = (h.astype(np.float64) / float(1 << 32) - 0.5).astype(np.float32).reshape(shape)
    if name.endswith("moving_variance:0") or name.endswith("gamma:0"):
        v = v + np.float32(1.0)
    return v


def _expected(layers):
    from lidbox_amd.models.hdf5_reader import _VAR_SUFFIX
    out = {}
    for _, weights in layers:
        for wname, shape in weights:
            parts = wname.split("/")
            out[parts[-2] + _VAR_SUFFIX[parts[-1].split(":")[0]]] = _values(wname, shape)
    return out


@pytest.mark.parametrize("fname", ["keras_tdnn_weights.h5", "keras_tdnn_weights_latest.h5"])
def test_save_weights_layout_old_and_new_file_format(fname):
    """old-style groups (symbol tables, v1 object headers) and libver="latest" (superblock 3, v2 headers, link messages)"""
    import make_keras_h5 as gen
    from lidbox_amd.models import hdf5_reader as h5
    path = os.path.join(GOLDEN, fname)
    f = h5.File(path)
    assert sorted(f.keys()) == sorted(n for n, _ in gen.TDNN_LAYERS)
    assert f.attrs["layer_names"].dtype.kind == "S"                                   # fixed-length strings (h5py 2.x era files)
    assert h5._as_str_list(f.attrs["layer_names"]) == [n for n, _ in gen.TDNN_LAYERS]
    assert f.attrs["backend"] in ("tensorflow", b"tensorflow")
    assert list(np.asarray(f["input"].attrs["weight_names"]).shape) == [0]          # layers without weights
    d = f["frame1/frame1/kernel:0"]
    assert d.shape == (5, 6, 8) and d.dtype == np.dtype("<f4")
    assert np.array_equal(d[1, 2, :], _values("frame1/kernel:0", (5, 6, 8))[1, 2, :])
    got = h5.load_keras_weights(path)
    want = _expected(gen.TDNN_LAYERS)
    assert set(got) == set(want) == {"frame1.W", "frame1.b", "frame2.W", "frame2.b", "segment1.W", "segment1.b", "outputs.W", "outputs.b"}
    for k in want:
        assert got[k].dtype == np.float32 and np.array_equal(got[k], want[k]), k
    assert [p for p, _ in f.visit_datasets()][0] == "frame1/frame1/bias:0"
    with pytest.raises(KeyError):
        f["frame1/nope"]


def test_full_model_checkpoint_layout_with_nested_batchnorm_layers():
    """model.save / ModelCheckpoint layout: model_weights + optimizer_weights + JSON attributes (variable-length strings)"""
    import json
    import make_keras_h5 as gen
    from lidbox_amd.models import hdf5_reader as h5
    path = os.path.join(GOLDEN, "keras_frontend_checkpoint.hdf5")
    f = h5.File(path)
    assert sorted(f.keys()) == ["model_weights", "optimizer_weights"]
    cfg = json.loads(f.attrs["model_config"])
    assert cfg["config"]["name"] == "x-vector-2D" and len(cfg["config"]["layers"]) == len(gen.FRONTEND_LAYERS)
    assert json.loads(f.attrs["training_config"])["loss"] == "sparse_categorical_crossentropy"
    assert f["model_weights"].attrs["layer_names"].dtype == object                    # variable-length strings (global heap)
    assert int(f["optimizer_weights/Adam/iter:0"].read()) == 17
    got = h5.load_keras_weights(path)
    want = _expected(gen.FRONTEND_LAYERS)
    assert set(got) == set(want) and "frame2d_1_bn.moving_variance" in got and "frame2d_2_conv.W" in got
    for k in want:
        assert np.array_equal(got[k], want[k]), k
    assert got["frame2d_2_conv.W"].shape == (1, 3, 4, 2) and float(got["frame2d_1_bn.gamma"].min()) >= 0.5


def test_reader_rejects_what_it_does_not_understand(tmp_path):
    from lidbox_amd.models import hdf5_reader as h5
    p = tmp_path / "x.h5"
    p.write_bytes(b"not an hdf5 file at all" * 10)
    with pytest.raises(h5.Hdf5FormatError):
        h5.File(str(p))
    raw = open(os.path.join(GOLDEN, "keras_tdnn_weights.h5"), "rb").read()
    q = tmp_path / "truncated.h5"
    q.write_bytes(raw[:len(raw) // 3])
    with pytest.raises((h5.Hdf5FormatError, ValueError, IndexError)):
        h5.load_keras_weights(str(q))
    # a valid HDF5 file that is not a Keras weight file
    with pytest.raises(h5.Hdf5FormatError):
        f = h5.File(os.path.join(GOLDEN, "keras_frontend_checkpoint.hdf5"))
        assert "layer_names" not in f["optimizer_weights"].attrs
        raise h5.Hdf5FormatError("optimizer_weights has no layer_names")


def test_checkpoint_name_rule_covers_reference_hdf5_files(tmp_path):
    """reference keras_utils.py:41-42, 103-121 on `epoch{epoch:06d}__val_loss{val_loss:.12f}.hdf5` names"""
    import shutil
    from lidbox_amd.models import keras_utils as ku
    src = os.path.join(GOLDEN, "keras_tdnn_weights.h5")
    names = ["epoch000001__val_loss1.500000000000.hdf5", "epoch000002__val_loss0.250000000000.hdf5",
             "epoch000003__val_loss0.750000000000.hdf5"]
    for n in names:
        shutil.copy(src, tmp_path / n)
    assert ku.parse_checkpoint_value(str(tmp_path / names[1]), "epoch") == "000002"
    assert ku.parse_checkpoint_value(str(tmp_path / names[1]), "val_loss") == "0.250000000000"
    assert os.path.basename(ku.KerasWrapper.get_best_checkpoint_path(str(tmp_path), key="val_loss", mode="min")) == names[1]
    assert os.path.basename(ku.KerasWrapper.get_best_checkpoint_path(str(tmp_path))) == names[2]
    w = ku.read_weights_file(str(tmp_path / names[0]))
    assert w["segment1.W"].shape == (16, 10)
