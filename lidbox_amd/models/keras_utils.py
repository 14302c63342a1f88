"""
Counterpart of lidbox/models/keras_utils.py (reference :1-214): the config-driven wrapper that is the direct
CALLER of the hot path -- `KerasWrapper.from_config(config)` builds model + optimizer + loss + metrics from the
lidbox experiment config, `fit` runs the train step over a dataset, checkpoints are chosen by the reference's file
name rule (`epoch{epoch:06d}__val_loss{val_loss:.12f}`), and `from_config_as_embedding_extractor_fn` turns the best
checkpoint into the extractor that `lidbox_amd.data.steps.extract_embeddings` consumes.

What differs from the reference, by necessity:
  * the model is a `lidbox_amd.models.*` engine and every step is `lidbox_amd.train.Trainer.train_step` (hipGraph
    replay); datasets are iterables of `(inputs [B,T,C], targets [B])` batches (what `as_supervised` + `batch` give);
  * optimizer: Adam (the train step's fused Keras-Adam kernel); loss: SparseCategoricalCrossentropy(from_logits=True)
    or `lidbox.losses.SparseAngularProximity`; metrics: (Sparse)AverageDetectionCost, SparseCategoricalAccuracy;
  * checkpoints WRITTEN here are `.npz` files of Keras-layout arrays (`get_weights()`); the reference's own Keras HDF5
    checkpoints (`epoch...hdf5`, keras_utils.py:57-64) are READ through `hdf5_reader` (no h5py in this build), so
    `load_weights` and the embedding-extractor configs accept a checkpoint directory the reference produced;
  * callbacks: ModelCheckpoint, EarlyStopping, LearningRateDateLogger; TensorBoard is accepted and ignored.
"""
import datetime
import importlib
import io
import os
import sys
import time

import numpy as np
import torch

from .. import _native as nv
from .. import metrics as lidbox_metrics
from ..losses import SparseAngularProximity
from ..train import Trainer

MODELS_IMPORT_PATH = "lidbox_amd.models."
CHECKPOINT_SUFFIX = ".npz"
KERAS_HDF5_SUFFIXES = (".hdf5", ".h5")          # what the reference's ModelCheckpoint / save_weights write


def read_weights_file(path):
    """{parameter name: numpy array in the Keras layout} from a checkpoint of this build (.npz) or a Keras HDF5 weight /
    model file written by the reference (keras_utils.py:57-64: `epoch{epoch:06d}__val_loss{val_loss:.12f}.hdf5`)."""
    if path.endswith(KERAS_HDF5_SUFFIXES):
        from .hdf5_reader import load_keras_weights
        return load_keras_weights(path)
    return dict(np.load(path))


def _set_weights_checked(model, weights, path):
    """every parameter of the model must be in the file with the model's shape (Keras `load_weights` raises likewise);
    entries the model has no use for (variables of layers outside the hot path saved next to it) are ignored with a warning"""
    want = dict(list(model.layout.items()) + list(getattr(model, "state_layout", {}).items()))
    missing = sorted(set(want) - set(weights))
    extra = sorted(set(weights) - set(want))
    if missing:
        raise ValueError("%s does not match the model: missing %s (unexpected %s)" % (path, missing[:6], extra[:6]))
    if extra:
        import warnings
        warnings.warn("%s: ignoring %d entries the model does not have (%s ...)" % (path, len(extra), extra[:3]))
        weights = {k: v for k, v in weights.items() if k in want}
    for name, (_, shape) in want.items():
        if tuple(np.shape(weights[name])) != tuple(shape):
            raise ValueError("%s: %s has shape %s, the model expects %s" % (path, name, np.shape(weights[name]), tuple(shape)))
    model.set_weights(weights)


def experiment_cache_from_config(config):
    """reference keras_utils.py:19-24"""
    experiment_config = config["sklearn_experiment"] if "sklearn_experiment" in config else config["experiment"]
    return os.path.join(experiment_config["cache_directory"], experiment_config["model"]["key"], experiment_config["name"])


def best_model_checkpoint_from_config(config):
    """reference keras_utils.py:27-38"""
    checkpoint_callbacks = [d for d in config["experiment"].get("callbacks", []) if d["cls"] == "ModelCheckpoint"]
    checkpoint_kwargs = checkpoint_callbacks[0].get("kwargs", {}) if checkpoint_callbacks else {}
    if "filepath" in checkpoint_kwargs:
        checkpoints_dir = os.path.dirname(checkpoint_kwargs["filepath"])
    else:
        checkpoints_dir = os.path.join(experiment_cache_from_config(config), "checkpoints")
    return KerasWrapper.get_best_checkpoint_path(checkpoints_dir, key=checkpoint_kwargs.get("monitor"),
                                                 mode=checkpoint_kwargs.get("mode"))


def parse_checkpoint_value(checkpoint_path, key):
    """reference keras_utils.py:41-42 (with this build's checkpoint suffix)"""
    value = checkpoint_path.split(key)[-1].split("__")[0]
    for suffix in (CHECKPOINT_SUFFIX,) + KERAS_HDF5_SUFFIXES:
        value = value.split(suffix)[0]
    return value


class SparseCategoricalAccuracy:
    """tf.keras.metrics.SparseCategoricalAccuracy: streaming mean of argmax(pred) == target"""
    name = "sparse_categorical_accuracy"

    def __init__(self, **kwargs):
        self.reset_states()

    def reset_states(self):
        self.correct, self.count = 0, 0

    def update_state(self, y_true, y_pred):
        self.correct += int((y_pred.argmax(dim=1) == torch.as_tensor(y_true, device=y_pred.device).long()).sum())
        self.count += int(y_pred.shape[0])

    def result(self):
        return self.correct / max(1, self.count)


def init_metric_from_config(config):
    """reference keras_utils.py:45-52"""
    if config["cls"].endswith("AverageDetectionCost"):
        args = [config["threshold_linspace"][k] for k in ("start", "stop", "num")]
        thresholds = np.linspace(*args)
        return getattr(lidbox_metrics, config["cls"])(config["N"], thresholds)
    if config["cls"] == "SparseCategoricalAccuracy":
        return SparseCategoricalAccuracy(**config.get("kwargs", {}))
    raise ValueError("unsupported metric %r" % (config["cls"],))


class ModelCheckpoint:
    """tf.keras.callbacks.ModelCheckpoint subset: filepath format, monitor, mode, save_best_only"""

    def __init__(self, filepath, monitor="val_loss", mode="min", save_best_only=False, **unused):
        self.filepath, self.monitor, self.mode, self.save_best_only = filepath, monitor, mode, save_best_only
        self.best = None

    def on_epoch_end(self, wrapper, epoch, logs):
        value = logs.get(self.monitor)
        better = self.best is None or value is None or (value < self.best if self.mode != "max" else value > self.best)
        if self.save_best_only and not better:
            return
        if better and value is not None:
            self.best = value
        path = self.filepath.format(epoch=epoch + 1, **logs)
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        np.savez(path, **wrapper.keras_model.get_weights())


class EarlyStopping:
    """tf.keras.callbacks.EarlyStopping subset: monitor, patience, mode, min_delta"""

    def __init__(self, monitor="val_loss", patience=0, mode="min", min_delta=0.0, verbose=0, **unsupported):
        # arguments that would change behaviour but are not implemented must not be swallowed
        bad = {k: v for k, v in unsupported.items() if not (k == "restore_best_weights" and not v) and not (k == "baseline" and v is None)}
        if bad:
            raise ValueError("EarlyStopping: unsupported arguments %s" % sorted(bad))
        self.monitor, self.patience, self.mode, self.min_delta = monitor, int(patience), mode, float(min_delta)
        self.best, self.wait = None, 0

    def on_epoch_end(self, wrapper, epoch, logs):
        value = logs.get(self.monitor)
        if value is None:
            return
        improved = self.best is None or (value < self.best - self.min_delta if self.mode != "max"
                                         else value > self.best + self.min_delta)
        if improved:
            self.best, self.wait = value, 0
        else:
            self.wait += 1
            if self.wait >= self.patience:                   # tf.keras: stops once `patience` epochs brought no improvement
                wrapper.stop_training = True


class LearningRateDateLogger:
    """reference keras_utils.py:82-95"""

    def __init__(self, output_stream=sys.stdout, **kwargs):
        self.output_stream = output_stream

    def on_epoch_begin(self, wrapper, epoch):
        print(str(datetime.datetime.now()), "-", "Adam", "learning rate:", wrapper.trainer.opt["lr"], file=self.output_stream)


class _Ignored:
    def __init__(self, **kwargs):
        pass


_CALLBACKS = {"ModelCheckpoint": ModelCheckpoint, "EarlyStopping": EarlyStopping,
              "LearningRateDateLogger": LearningRateDateLogger, "TensorBoard": _Ignored}


def init_callback_from_config(config, cache_dir):
    """reference keras_utils.py:55-79"""
    user_kwargs = dict(config.get("kwargs", {}))
    if config["cls"] == "ModelCheckpoint":
        default_checkpoint_format = "epoch{epoch:06d}__val_loss{val_loss:.12f}" + CHECKPOINT_SUFFIX
        callback_kwargs = {"filepath": os.path.join(os.path.join(cache_dir, "checkpoints"),
                                                    config.get("format", default_checkpoint_format))}
        callback_kwargs.update(user_kwargs)
        if not callback_kwargs["filepath"].endswith(CHECKPOINT_SUFFIX):
            callback_kwargs["filepath"] = os.path.splitext(callback_kwargs["filepath"])[0] + CHECKPOINT_SUFFIX
        os.makedirs(os.path.dirname(callback_kwargs["filepath"]), exist_ok=True)
    else:
        callback_kwargs = user_kwargs
    if config["cls"] not in _CALLBACKS:
        raise ValueError("unsupported callback %r" % (config["cls"],))
    return _CALLBACKS[config["cls"]](**callback_kwargs)


def _loss_from_config(loss_conf):
    """reference keras_utils.py:141-142: `getattr(tf.keras.losses, cls)(**kwargs)`"""
    cls, kwargs = loss_conf["cls"], loss_conf.get("kwargs", {})
    if cls == "SparseCategoricalCrossentropy":
        # from_logits=True: the models' default log-softmax outputs (softmax of log-softmax is the softmax).
        # from_logits=False (the Keras default): probabilities, i.e. a model built with output_activation="softmax"
        # (cnn.py:43-44) -- Trainer checks that the model and the loss agree
        return "sparse_categorical_crossentropy" if kwargs.get("from_logits", False) else "sparse_categorical_crossentropy_probs"
    if cls == "SparseAngularProximity":
        return SparseAngularProximity(**kwargs)
    raise ValueError("unsupported loss %r" % (cls,))


def lr_schedule_from_config(conf):
    """reference keras_utils.py:137-139: `getattr(tf.keras.optimizers.schedules, cls)(**kwargs)` as a plain function of the
    optimizer step (Keras' `iterations`: 0 at the first update).  ExponentialDecay and PiecewiseConstantDecay with their
    Keras argument names and semantics; float32 arithmetic as the Keras ops."""
    cls, kw = conf["cls"], dict(conf.get("kwargs", {}))
    if cls == "ExponentialDecay":
        lr0, steps, rate = np.float32(kw["initial_learning_rate"]), np.float32(kw["decay_steps"]), np.float32(kw["decay_rate"])
        staircase = bool(kw.get("staircase", False))

        def sched(step):
            p = np.float32(step) / steps
            if staircase:
                p = np.floor(p)
            return float(lr0 * np.power(rate, p, dtype=np.float32))
        return sched
    if cls == "PiecewiseConstantDecay":
        boundaries, values = list(kw["boundaries"]), [float(v) for v in kw["values"]]
        if len(values) != len(boundaries) + 1:
            raise ValueError("PiecewiseConstantDecay: len(values) must be len(boundaries) + 1")

        def sched(step):
            for b, v in zip(boundaries, values):
                if step <= b:
                    return v
            return values[-1]
        return sched
    raise ValueError("unsupported learning-rate schedule %r (ExponentialDecay, PiecewiseConstantDecay)" % (cls,))


def _optimizer_from_config(opt_conf):
    """reference keras_utils.py:135-140: `getattr(tf.keras.optimizers, cls)(**kwargs)` -- Adam, SGD and RMSprop with their Keras
    argument names; `lr_scheduler` as there"""
    names = {"Adam": ("beta_1", "beta_2", "epsilon"), "SGD": ("momentum", "nesterov"), "RMSprop": ("rho", "momentum", "epsilon", "centered")}
    if opt_conf["cls"] not in names:
        raise ValueError("unsupported optimizer %r (Adam, SGD, RMSprop)" % (opt_conf["cls"],))
    kw = dict(opt_conf.get("kwargs", {}))
    out = {"cls": opt_conf["cls"]}
    if "lr_scheduler" in kw:
        out["lr_schedule"] = lr_schedule_from_config(kw.pop("lr_scheduler"))
    for src in ("learning_rate", "lr"):
        if src in kw:
            out["lr"] = float(kw.pop(src))
    for name in names[opt_conf["cls"]]:
        if name in kw:
            v = kw.pop(name)
            out[name] = bool(v) if name in ("nesterov", "centered") else float(v)
    kw.pop("name", None)
    if kw:
        raise ValueError("%s: unsupported optimizer arguments %s" % (opt_conf["cls"], sorted(kw)))
    return out


class KerasWrapper:
    """reference keras_utils.py:98-214"""

    @staticmethod
    def get_best_checkpoint_path(checkpoints_dir, key=None, mode=None):
        """reference keras_utils.py:103-121"""
        if key is None:
            key = "epoch"
        checkpoints = [p.path for p in os.scandir(checkpoints_dir)
                       if p.is_file() and p.name.endswith((CHECKPOINT_SUFFIX,) + KERAS_HDF5_SUFFIXES)]
        key_fn = lambda p: parse_checkpoint_value(p, key)       # noqa: E731
        best_path = None
        if checkpoints:
            if key == "epoch":
                best_path = max(checkpoints, key=lambda p: int(key_fn(p)))
            else:
                assert mode in ("min", "max"), mode
                best_path = (min if mode == "min" else max)(checkpoints, key=lambda p: float(key_fn(p)))
        return best_path

    @classmethod
    def get_model_filepath(cls, basedir, model_key):
        return os.path.join(basedir, cls.__name__.lower() + '-' + model_key)

    @staticmethod
    def _load_model(model_conf, input_shape, output_shape):
        model_module = importlib.import_module(MODELS_IMPORT_PATH + model_conf["key"])
        num_outputs = int(np.squeeze(np.asarray(output_shape)))
        return model_module, model_module.loader(tuple(input_shape), num_outputs, **model_conf.get("kwargs", {}))

    @classmethod
    def from_config(cls, config):
        """reference keras_utils.py:127-149"""
        experiment_cache = experiment_cache_from_config(config)
        os.makedirs(experiment_cache, exist_ok=True)
        exp = config["experiment"]
        _, model = cls._load_model(exp["model"], exp["input_shape"], exp["output_shape"])
        loss = _loss_from_config(exp["loss"])
        if isinstance(loss, SparseAngularProximity) and model.output_activation is not None:
            raise ValueError("SparseAngularProximity needs a model without output activation")
        metrics = [init_metric_from_config(c) for c in exp.get("metrics", [])]
        callbacks = [init_callback_from_config(c, experiment_cache) for c in exp.get("callbacks", [])]
        return cls(model, exp["model"]["key"], callbacks, loss=loss, optimizer=_optimizer_from_config(exp["optimizer"]),
                   metrics=metrics)

    @classmethod
    def from_config_as_embedding_extractor_fn(cls, config):
        """reference keras_utils.py:151-173: best checkpoint -> `as_embedding_extractor` callable [B,T,C] -> [B,D]"""
        experiment_cache = experiment_cache_from_config({"experiment": {
            "cache_directory": config["cache_directory"], "model": config["model"], "name": config["experiment_name"]}})
        model_module, model = cls._load_model(config["model"], config["input_shape"], config["output_shape"])
        path = cls.get_best_checkpoint_path(os.path.join(experiment_cache, "checkpoints"),
                                            key=config["best_checkpoint"]["monitor"], mode=config["best_checkpoint"]["mode"])
        if path is None:
            raise FileNotFoundError("no checkpoint under %s" % os.path.join(experiment_cache, "checkpoints"))
        _set_weights_checked(model, read_weights_file(path), path)
        return getattr(model_module, "as_embedding_extractor")(model)

    def __init__(self, keras_model, model_key, callbacks, loss="sparse_categorical_crossentropy", optimizer=None,
                 metrics=None):
        self.model_key = model_key
        self.keras_model = keras_model
        self.initial_epoch = 0
        self.callbacks = callbacks
        self.loss = loss
        self.metrics = list(metrics or [])
        self.trainer = Trainer(keras_model, loss=loss, optimizer=optimizer)
        self.stop_training = False
        self._staging = {}

    def to_disk(self, basedir):
        """reference keras_utils.py:181-184 (weights in Keras layouts, .npz)"""
        model_path = self.get_model_filepath(basedir, self.model_key) + CHECKPOINT_SUFFIX
        os.makedirs(basedir, exist_ok=True)
        np.savez(model_path, **self.keras_model.get_weights())
        return model_path

    def load_weights(self, path):
        """reference keras_utils.py:186-188"""
        self.initial_epoch = int(parse_checkpoint_value(path, key="epoch"))
        _set_weights_checked(self.keras_model, read_weights_file(path), path)

    # ------------------------------------------------------------------ fit / evaluate
    def _stage(self, x, y):
        """copy a batch into buffers that keep their address per shape: the captured step graph is keyed on them"""
        x = torch.as_tensor(x, dtype=torch.float32)
        y = torch.as_tensor(y).to(torch.int32).reshape(-1)
        key = tuple(x.shape)
        buf = self._staging.pop(key, None)
        if buf is None:
            # at most 4 shapes stay staged (what Trainer / the model keep graphs and workspaces for): `fit` expects a
            # small set of fixed batch shapes -- bucket or pad variable-length batches (group_by_input_length) up front
            while len(self._staging) >= 4:
                self._staging.pop(next(iter(self._staging)))
            dev = self.keras_model.device
            buf = (torch.empty(key, dtype=torch.float32, device=dev), torch.empty(key[0], dtype=torch.int32, device=dev))
        self._staging[key] = buf                             # most recently used last
        buf[0].copy_(x)
        buf[1].copy_(y)
        return buf

    def _eval_loss(self, out, y):
        """(mean loss of the batch, the scores the metrics read) on the library's own kernels: tf.math.l2_normalize + the loss for
        SparseAngularProximity (reference losses.py:25-52), sparse categorical cross-entropy on the model's log-probabilities /
        probabilities otherwise (keras_utils.py:141-147)"""
        out = out.contiguous()
        B, N = out.shape
        st = nv.current_stream()
        with torch.cuda.device(out.device):
            if isinstance(self.loss, SparseAngularProximity):
                zn = torch.empty_like(out)
                nv.check(nv.lib.lidbox_l2_normalize_fwd(nv.ptr(out), B, N, nv.ptr(zn), st))
                return float(self.loss(y, zn)), self.loss.predict(zn)
            loss = torch.empty(1, dtype=torch.float32, device=out.device)
            y32 = y.to(torch.int32).contiguous()
            if getattr(self.keras_model, "output_activation", None) == "softmax":
                # probabilities (a model built with output_activation="softmax", from_logits=False): Keras clips to [1e-7, 1 - 1e-7],
                # renormalises and takes -log; the one configuration of this method that still prepares its operand with tensor ops
                q = out.clamp(1e-7, 1 - 1e-7)
                logp = (torch.log(q) - torch.log(q.sum(dim=1, keepdim=True))).contiguous()
                nv.check(nv.lib.lidbox_nll_fwd_bwd(nv.ptr(logp), nv.ptr(y32), B, N, 1.0 / B, nv.ptr(loss), None, st))
            else:
                nv.check(nv.lib.lidbox_nll_fwd_bwd(nv.ptr(out), nv.ptr(y32), B, N, 1.0 / B, nv.ptr(loss), None, st))
        return float(loss), out

    def evaluate(self, dataset):
        """mean loss and metric values over a dataset of (inputs, targets) batches (Keras `Model.evaluate`)"""
        for m in self.metrics:
            m.reset_states()
        total, count = 0.0, 0
        for x, y in dataset:
            xs, ys = self._stage(x, y)
            out = self.keras_model(xs, training=False)
            loss, scores = self._eval_loss(out, ys)
            total += loss * xs.shape[0]
            count += xs.shape[0]
            for m in self.metrics:
                if isinstance(m, lidbox_metrics.AverageDetectionCost):
                    m._update_sparse(ys, scores)
                else:
                    m.update_state(ys, scores)
        logs = {"loss": total / max(1, count)}
        for m in self.metrics:
            logs[getattr(m, "name", type(m).__name__)] = float(m.result())
        return logs

    def fit(self, training_dataset, validation_dataset, user_kwargs):
        """reference keras_utils.py:190-203.  Datasets are re-iterable collections of (inputs, targets) batches;
        user_kwargs: epochs, steps_per_epoch, validation_freq, verbose.  Returns {"history": {...}, "epoch": [...]}"""
        kwargs = {"shuffle": False, "validation_freq": 1, "verbose": 2}
        kwargs.update(user_kwargs)
        epochs = int(kwargs.get("epochs", 1))
        steps_per_epoch = kwargs.get("steps_per_epoch")
        history = {"history": {}, "epoch": []}
        self.stop_training = False
        for epoch in range(self.initial_epoch, epochs):
            for cb in self.callbacks:
                if hasattr(cb, "on_epoch_begin"):
                    cb.on_epoch_begin(self, epoch)
            t0 = time.time()
            losses, n = [], 0
            for step, (x, y) in enumerate(training_dataset):
                if steps_per_epoch is not None and step >= int(steps_per_epoch):
                    break
                xs, ys = self._stage(x, y)
                # data parallelism: the global batch of this step, exchanged on every rank at every step (rank-symmetric; the
                # shards of a last, smaller batch need not shrink together)
                gb = self.trainer.global_batch_of(xs.shape[0]) if self.trainer.sync.active else None
                losses.append((self.trainer.train_step(xs, ys, global_batch=gb).clone(), xs.shape[0]))
                n += xs.shape[0]
            logs = {"loss": float(sum(float(l) * b for l, b in losses) / max(1, n))}
            self.trainer.sync_state()        # data parallelism: the replicas' BatchNormalization running statistics, averaged before they are used
            if validation_dataset is not None and (epoch + 1) % int(kwargs["validation_freq"]) == 0:
                logs.update({"val_" + k: v for k, v in self.evaluate(validation_dataset).items()})
            if kwargs["verbose"]:
                print("Epoch %d/%d - %.1fs - %s" % (epoch + 1, epochs, time.time() - t0,
                                                     " - ".join("%s: %.6f" % kv for kv in logs.items())))
            history["epoch"].append(epoch)
            for k, v in logs.items():
                history["history"].setdefault(k, []).append(v)
            for cb in self.callbacks:
                if hasattr(cb, "on_epoch_end"):
                    cb.on_epoch_end(self, epoch, logs)
            if self.stop_training:
                break
        return history

    def count_params(self):
        return self.keras_model.count_params()

    def __str__(self):
        with io.StringIO() as sstream:
            m = self.keras_model
            print('Model: "%s"' % m.name, file=sstream)
            for name, (off, shape) in m.layout.items():
                print("  %-16s %-20s %d" % (name, tuple(shape), int(np.prod(shape))), file=sstream)
            print("Total params: %d" % m.count_params(), file=sstream)
            return sstream.getvalue()
