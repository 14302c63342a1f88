"""
Counterpart of the ONE lidbox.data.steps entry that sits on the hot path:
`extract_features(ds, config)` (reference lidbox/data/steps.py:708-736) with the same config
schema (`_feature_extraction_kwargs_to_args`, steps.py:94-104):

    {"type": "logmelspectrogram" | "mfcc" | "melspectrogram" | "spectrogram" | "db_spectrogram",
     "spectrogram": {...}, "melspectrogram": {...}, "mfcc": {...}, "db_spectrogram": {...},
     "sample_minmax_scaling": {...}, "window_normalization": {...},
     "batch_size": 1 | "group_by_input_length": {"max_batch_size": n}, "device": ...}

The reference maps this over a tf.data.Dataset (batch -> map -> unbatch); here `ds` is any
iterable of element dicts holding at least `signal` (1-D float32 tensor or array) and
`sample_rate`, and the result is a generator of the same dicts with `input` ([T, C] tensor on the
HIP device) and `feature_type` added.  Everything else in lidbox.data.steps (tf.data plumbing,
augmentation, caching, TensorBoard dumps) is out of scope.
"""
import logging

import torch

from . import tf_utils

logger = logging.getLogger("lidbox_amd")

VALID_FEATURE_ARGS = ["type", "spectrogram", "melspectrogram", "mfcc", "db_spectrogram", "sample_minmax_scaling",
                      "window_normalization"]


def _feature_extraction_kwargs_to_args(config):
    """reference steps.py:94-104"""
    return [config.get(arg, {}) for arg in VALID_FEATURE_ARGS]


def _get_device_or_default(config):
    """reference steps.py:115-122 -- there is no CPU path here: the default IS the HIP device."""
    dev = config.get("device", "cuda")
    if isinstance(dev, str):
        name = dev.upper().lstrip("/")
        if name.startswith("GPU"):
            dev = "cuda" + (":" + name[4:] if name[3:4] == ":" and name[4:] else "")
        elif name.startswith("CPU"):
            raise ValueError("lidbox_amd extracts features on the HIP device only (got device=%r)" % (config.get("device"),))
    dev = torch.device(dev)
    if dev.type != "cuda":
        raise ValueError("lidbox_amd extracts features on the HIP device only (got device=%r)" % (config.get("device"),))
    return dev


def _batches(ds, config):
    if "group_by_input_length" in config:
        # reference steps.py:725-728 + group_by_axis_length (:751-773): same-length batches of bounded size
        max_bs = int(config["group_by_input_length"]["max_batch_size"])
        groups = {}
        for x in ds:
            n = int(torch.as_tensor(x["signal"]).shape[0])
            g = groups.setdefault(n, [])
            g.append(x)
            if len(g) == max_bs:
                yield groups.pop(n)
        for g in groups.values():
            if g:
                yield g
    else:
        bs = int(config.get("batch_size", 1))                                        # reference steps.py:730
        batch = []
        for x in ds:
            if batch and int(torch.as_tensor(x["signal"]).shape[0]) != int(torch.as_tensor(batch[0]["signal"]).shape[0]):
                raise ValueError("cannot batch signals of different lengths; use group_by_input_length")
            batch.append(x)
            if len(batch) == bs:
                yield batch
                batch = []
        if batch:
            yield batch


def extract_features(ds, config):
    """reference lidbox/data/steps.py:708-736"""
    feature_type = config["type"]
    args = _feature_extraction_kwargs_to_args(config)
    device = _get_device_or_default(config)
    logger.info("Extracting '%s' features on device '%s' with arguments:\n  %s", feature_type, device,
                "\n  ".join(repr(a) for a in args[1:]))
    for batch in _batches(ds, config):
        signals = torch.stack([torch.as_tensor(x["signal"], dtype=torch.float32) for x in batch]).to(device)
        rates = [int(x["sample_rate"]) for x in batch]
        feats = tf_utils.extract_features(signals, rates, *args)
        for i, x in enumerate(batch):                                                # unbatch (:736)
            yield dict(x, input=feats[i], feature_type=feature_type)
