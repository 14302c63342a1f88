"""Convert a Keras HDF5 weight / checkpoint file of the reference (lidbox/models/keras_utils.py:57-64) into the .npz
checkpoint format of this build, keeping the file name rule (epoch / val_loss are parsed from the name).
usage: python tools/keras_h5_to_npz.py checkpoints/epoch000003__val_loss0.123456789012.hdf5 [out_dir]
Needs no h5py / TensorFlow (lidbox_amd.models.hdf5_reader parses the file)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidbox_amd.models.hdf5_reader import load_keras_weights  # noqa: E402


def main(argv):
    if not argv:
        raise SystemExit(__doc__)
    src = argv[0]
    out_dir = argv[1] if len(argv) > 1 else os.path.dirname(os.path.abspath(src))
    weights = load_keras_weights(src)
    dst = os.path.join(out_dir, os.path.splitext(os.path.basename(src))[0] + ".npz")
    np.savez(dst, **weights)
    print(dst, "%d arrays, %d parameters" % (len(weights), sum(v.size for v in weights.values())))


if __name__ == "__main__":
    main(sys.argv[1:])
