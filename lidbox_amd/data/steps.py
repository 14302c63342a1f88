"""
Counterpart of the lidbox.data.steps entries on and next to the hot path.  `ds` is any iterable of element
dicts (the reference maps the same functions over a tf.data.Dataset); results are generators of dicts.

  extract_features      reference steps.py:708-736   (the hot path's boundary)
  compute_rms_vad       reference steps.py:417-432   energy VAD decisions             
  apply_vad             reference steps.py:183-200   drop the non-speech frames        
  create_signal_chunks  reference steps.py:579-632   fixed-length chunks, new ids      
  extract_embeddings    reference steps.py:674-705   batched embedding extraction      

The signal steps gather `launch_batch` elements into one ragged device batch per kernel launch
(lidbox_amd/features/signal_ops.py); the order of elements is preserved.

`extract_features(ds, config)` takes the reference's config
schema (`_feature_extraction_kwargs_to_args`, steps.py:94-104):

    {"type": "logmelspectrogram" | "mfcc" | "melspectrogram" | "spectrogram" | "db_spectrogram",
     "spectrogram": {...}, "melspectrogram": {...}, "mfcc": {...}, "db_spectrogram": {...},
     "sample_minmax_scaling": {...}, "window_normalization": {...},
     "batch_size": 1 | "group_by_input_length": {"max_batch_size": n}, "device": ...}

The reference maps this over a tf.data.Dataset (batch -> map -> unbatch); here `ds` is any
iterable of element dicts holding at least `signal` (1-D float32 tensor or array) and
`sample_rate`, and the result is a generator of the same dicts with `input` ([T, C] tensor on the
HIP device) and `feature_type` added.  Everything else in lidbox.data.steps (tf.data plumbing,
file-based augmentation, caching, TensorBoard dumps) is out of scope.
"""
import logging

import numpy as np
import torch

from ..features import audio as audio_features
from ..features import signal_ops
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
        sigs = [torch.as_tensor(x["signal"]) for x in batch]
        if not all(t.dtype == torch.int16 for t in sigs):                            # 16-bit PCM stays int16: the kernel reads it in place
            sigs = [t.to(torch.float32) for t in sigs]
        signals = torch.stack(sigs).to(device)
        rates = [int(x["sample_rate"]) for x in batch]
        feats = tf_utils.extract_features(signals, rates, *args)
        for x, f in zip(batch, feats.unbind(0)):                                     # unbatch (:736): one call for all the views
            yield dict(x, input=f, feature_type=feature_type)


# ------------------------------------------------------------------ signal steps
def _launch_batches(ds, launch_batch):
    batch = []
    for x in ds:
        batch.append(x)
        if len(batch) == launch_batch:
            yield batch
            batch = []
    if batch:
        yield batch


def _vad_frame_length(sample_rate, vad_frame_length_ms):
    """reference steps.py:192-193: int32(float32(sr) * (1e-3 * float32(ms)))"""
    sec = np.float32(1e-3) * np.float32(vad_frame_length_ms)
    return int(np.float32(sample_rate) * sec)


def compute_rms_vad(ds, strength, vad_frame_length_ms, min_non_speech_length_ms=0, launch_batch=256):
    """reference steps.py:417-432: adds `vad_is_speech` (bool [num_frames], on the HIP device) and
    `vad_frame_length_ms` to every element."""
    logger.info("Computing voice activity detection decisions by mean RMS values on %d ms long windows.\n"
                "Minimum length of continuous non-speech segment before it is marked as non-speech is %d ms.",
                vad_frame_length_ms, min_non_speech_length_ms)
    for batch in _launch_batches(ds, launch_batch):
        by_rate = {}
        for i, x in enumerate(batch):
            by_rate.setdefault(int(x["sample_rate"]), []).append(i)
        decisions = [None] * len(batch)
        for rate, idx in by_rate.items():
            frame_step = audio_features.ms_to_frames(rate, vad_frame_length_ms)
            min_frames = int(audio_features.ms_to_frames(rate, min_non_speech_length_ms) / frame_step)
            r = signal_ops.RaggedSignals.from_list([batch[i]["signal"] for i in idx])
            vad = signal_ops.vad_decisions(r, frame_step, min_frames, strength)
            for i, d in zip(idx, signal_ops.split_frames(vad, vad["decisions"])):
                decisions[i] = d.to(torch.bool)
        for x, d in zip(batch, decisions):
            yield dict(x, vad_is_speech=d, vad_frame_length_ms=vad_frame_length_ms)


def apply_vad(ds, launch_batch=256):
    """reference steps.py:183-200: `signal` keeps only the frames whose `vad_is_speech` is set; the two VAD keys
    are dropped."""
    logger.info("Using previously computed voice activity decisions to drop signal frames marked as non-speech.")
    drop_keys_after_done = {"vad_frame_length_ms", "vad_is_speech"}
    for batch in _launch_batches(ds, launch_batch):
        by_len = {}
        for i, x in enumerate(batch):
            by_len.setdefault(_vad_frame_length(x["sample_rate"], x["vad_frame_length_ms"]), []).append(i)
        voiced = [None] * len(batch)
        for L, idx in by_len.items():
            r = signal_ops.RaggedSignals.from_list([batch[i]["signal"] for i in idx])
            dev = r.flat.device
            decs = [torch.as_tensor(batch[i]["vad_is_speech"]).to(device=dev, dtype=torch.uint8).reshape(-1) for i in idx]
            nf = r.lengths_host // L
            for i, d, n in zip(idx, decs, nf):
                if d.numel() != n:
                    raise ValueError("element %r: %d VAD decisions for %d frames" % (batch[i].get("id"), d.numel(), n))
            fo_h, fo_d = signal_ops._csr(nf, dev)
            dec = torch.cat(decs) if decs else torch.zeros(0, dtype=torch.uint8, device=dev)
            slots = torch.empty(int(fo_h[-1]), dtype=torch.int32, device=dev)
            counts = torch.zeros(r.B, dtype=torch.int32, device=dev)
            from .. import _native as nv
            with torch.cuda.device(dev):
                nv.check(nv.lib.lidbox_vad_scan(nv.ptr(dec), nv.ptr(fo_d), r.B, nv.ptr(slots), nv.ptr(counts),
                                                nv.current_stream()))
            out = signal_ops.apply_vad(r, dict(decisions=dec, slots=slots, counts=counts, frame_offsets_host=fo_h,
                                               frame_offsets=fo_d, frame_len=L))
            for i, v in zip(idx, out.split()):
                voiced[i] = v
        for x, v in zip(batch, voiced):
            yield {k: val for k, val in dict(x, signal=v).items() if k not in drop_keys_after_done}


def create_signal_chunks(ds, length_ms, step_ms, max_pad_ms=0, deterministic_output_order=True,
                         max_num_chunks_per_signal=int(1e6), avg_num_chunks_from_signals=100, launch_batch=256):
    """reference steps.py:579-632: every signal becomes its fixed-length chunks; `id` gets the 1-based chunk
    number appended (zero padded to round(log10(max_num_chunks_per_signal)) digits), `duration` is recomputed.
    Output order is the deterministic one (`deterministic_output_order` and the interleave block length only
    affect tf.data scheduling in the reference)."""
    logger.info("Dividing every signal in the dataset into new signals by creating signal chunks of length %d ms and "
                "offset %d ms. Maximum amount of padding allowed in the last chunk is %d ms.", length_ms, step_ms, max_pad_ms)
    id_str_padding = int(round(float(np.log10(np.float32(max_num_chunks_per_signal)))))              # steps.py:589
    for batch in _launch_batches(ds, launch_batch):
        by_rate = {}
        for i, x in enumerate(batch):
            by_rate.setdefault(int(x["sample_rate"]), []).append(i)
        chunks_of = [None] * len(batch)
        for rate, idx in by_rate.items():
            r = signal_ops.RaggedSignals.from_list([batch[i]["signal"] for i in idx])
            for n in r.lengths_host:
                L, S, _, _ = signal_ops.chunk_plan(n, rate, length_ms, step_ms, max_pad_ms)
                if max(0, 1 + (int(n) - L) // S) >= max_num_chunks_per_signal:                       # steps.py:608
                    raise ValueError("Too many chunks created from signal, cannot create unique utterance ids, raise "
                                     "the max_num_chunks_per_signal parameter")
            chunks, nch = signal_ops.signal_chunks(r, rate, length_ms, step_ms, max_pad_ms)
            c0 = 0
            for i, n in zip(idx, nch):
                chunks_of[i] = chunks[c0:c0 + int(n)]
                c0 += int(n)
        for x, ch in zip(batch, chunks_of):
            for k in range(ch.shape[0]):
                out = dict(x, signal=ch[k], id="%s-%s" % (x["id"], str(k + 1).zfill(id_str_padding)))
                if "duration" in x:
                    out["duration"] = float(np.float32(ch.shape[1] / int(x["sample_rate"])))          # steps.py:597
                yield out


# ------------------------------------------------------------------ embeddings
def extract_embeddings(ds, config):
    """reference steps.py:674-705.  config = {"extractors": [...], "batch_size": 1, "no_unbatch": False}.
    An extractor is, as in the reference, a model config + checkpoint description handed to
    `KerasWrapper.from_config_as_embedding_extractor_fn` (keys cache_directory, model, experiment_name, input_shape,
    output_shape, best_checkpoint; steps.py:680-681) -- or, additionally, an already built callable: the result of
    `module.as_embedding_extractor(model)`, or a model that has `.embed`, mapping inputs [B, T, C] to embeddings
    [B, D].  The embeddings of several extractors are concatenated on axis 1 (steps.py:693).  Elements of one batch
    must share the input shape (tf.data's `batch` has the same requirement)."""
    extractors = []
    for e in config["extractors"]:
        if isinstance(e, dict):
            from ..models.keras_utils import KerasWrapper
            fn = KerasWrapper.from_config_as_embedding_extractor_fn(e)
        else:
            fn = e.embed if hasattr(e, "embed") else e
        if not callable(fn):
            raise ValueError("extractors must be checkpoint configs or callables mapping inputs [B,T,C] to embeddings [B,D]")
        extractors.append(fn)
    logger.info("Using %d extractors", len(extractors))
    batch_size = int(config.get("batch_size", 1))
    logger.info("Batching inputs with batch size %s, extracting embeddings in batches.", batch_size)
    no_unbatch = bool(config.get("no_unbatch", False))
    for batch in _launch_batches(ds, batch_size):
        inputs = torch.stack([torch.as_tensor(x["input"], dtype=torch.float32) for x in batch])
        if not inputs.is_cuda:
            inputs = inputs.cuda()
        embeddings = torch.cat([fn(inputs) for fn in extractors], dim=1)                              # steps.py:693
        if no_unbatch:
            keys = batch[0].keys()
            out = {k: [x[k] for x in batch] for k in keys}
            out["input"], out["embedding"] = inputs, embeddings
            yield out
        else:
            for i, x in enumerate(batch):
                yield dict(x, embedding=embeddings[i])
