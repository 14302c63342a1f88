"""
Counterpart of the lidbox/util.py entries around prediction and scoring:

  predictions_to_dataframe, predict_with_model   reference util.py:17-38
  chunk_parent_id, merge_chunk_predictions       reference util.py:41-57   (chunk rows averaged on the device)
  classification_report                          reference util.py:60-105

`classification_report` keeps the reference's structure: sklearn computes the per-class report, the ROC-based
equal error rates and the confusion matrix on the host (as in the reference), C_avg runs on the HIP counting
kernel (`lidbox_amd.metrics.SparseAverageDetectionCost`).
"""
import numpy as np
import torch

from . import _native as nv
from . import metrics as lidbox_metrics


def predictions_to_dataframe(ids, predictions):
    """reference util.py:17-20"""
    import pandas as pd
    return (pd.DataFrame.from_dict({"id": ids, "prediction": predictions})
            .set_index("id", drop=True, verify_integrity=True)
            .sort_index())


def predict_with_model(model, ds, predict_fn=None):
    """reference util.py:23-38: `ds` yields BATCHES: dicts with `id` (sequence of str) and `input` ([B,T,C])."""
    if predict_fn is None:
        def predict_fn(x):
            return x["id"], model(x["input"], training=False)
    ids, predictions = [], []
    for x in ds:
        bid, pred = predict_fn(x)
        pred = pred.detach().cpu().numpy() if isinstance(pred, torch.Tensor) else np.asarray(pred)
        for i, p in zip(bid, pred):                                                  # unbatch, :33-35
            ids.append(i.decode("utf-8") if isinstance(i, bytes) else str(i))
            predictions.append(p)
    return predictions_to_dataframe(ids, predictions)


def chunk_parent_id(chunk_id):
    """reference util.py:41-42"""
    return chunk_id.rsplit('-', 1)[0]


def segment_mean(x, segment_offsets):
    """mean over consecutive row groups of x [rows, D] on the HIP device (lidbox_segment_mean)"""
    x = nv.require_gpu_tensor(x, "x", torch.float32).contiguous()
    off = torch.as_tensor(np.asarray(segment_offsets, np.int64)).to(x.device)
    nseg = off.numel() - 1
    out = torch.empty((nseg, x.shape[1]), dtype=torch.float32, device=x.device)
    for s0 in range(0, nseg, 65535):
        n = min(65535, nseg - s0)
        with torch.cuda.device(x.device):
            nv.check(nv.lib.lidbox_segment_mean(nv.ptr(x), nv.C.c_void_p(off.data_ptr() + 8 * s0), n, x.shape[1],
                                                nv.C.c_void_p(out.data_ptr() + 4 * s0 * x.shape[1]), nv.current_stream()))
    return out


def merge_chunk_predictions(chunk_predictions, merge_rows_fn=None):
    """reference util.py:47-57: group the chunk rows by parent id (sorted, as pandas groupby does) and merge them;
    the default merge (stack_and_average) runs on the device."""
    ids = [str(i) for i in chunk_predictions.index]
    parents = [chunk_parent_id(i) for i in ids]
    order = sorted(range(len(ids)), key=lambda k: (parents[k], k))          # groupby sorts by key, stable inside
    uniq, counts = [], []
    for k in order:
        if uniq and uniq[-1] == parents[k]:
            counts[-1] += 1
        else:
            uniq.append(parents[k])
            counts.append(1)
    offsets = np.concatenate(([0], np.cumsum(counts))).astype(np.int64)
    rows = [np.asarray(chunk_predictions.prediction.values[k]) for k in order]
    if merge_rows_fn is not None:
        merged = [merge_rows_fn(rows[int(a):int(b)]) for a, b in zip(offsets[:-1], offsets[1:])]
        return predictions_to_dataframe(uniq, merged)
    if not rows:
        return predictions_to_dataframe([], [])
    x = torch.from_numpy(np.stack(rows).astype(np.float32)).cuda()
    merged = segment_mean(x, offsets).cpu().numpy()
    return predictions_to_dataframe(uniq, list(merged))


def classification_report(true_sparse, pred_dense, label2target, dense2sparse_fn=None, num_cavg_thresholds=100):
    """reference util.py:60-105"""
    import sklearn.metrics
    true_sparse = np.asarray(true_sparse)
    pred_dense = np.asarray(pred_dense)
    if dense2sparse_fn is None:
        dense2sparse_fn = lambda pred: pred.argmax(axis=1)      # noqa: E731
    pred_sparse = dense2sparse_fn(pred_dense)
    report = sklearn.metrics.classification_report(
        true_sparse, pred_sparse, labels=list(range(len(label2target))), target_names=label2target,
        output_dict=True, zero_division=0)
    cavg_thresholds = np.linspace(pred_dense.min(), pred_dense.max(), num_cavg_thresholds)            # :76-79
    cavg = lidbox_metrics.SparseAverageDetectionCost(len(label2target), cavg_thresholds)
    cavg.update_state(torch.from_numpy(true_sparse.astype(np.int32)).cuda(),
                      torch.from_numpy(pred_dense.astype(np.float32)).cuda())
    report["avg_detection_cost"] = float(cavg.result())

    true_dense = np.eye(len(label2target))[true_sparse.astype(np.int64)]
    eer = np.zeros(len(label2target))
    for l, label in enumerate(label2target):                                                          # :92-96
        fpr, tpr, _ = sklearn.metrics.roc_curve(true_dense[:, l], pred_dense[:, l])
        fnr = 1 - tpr
        eer[l] = fpr[np.nanargmin(np.absolute(fnr - fpr))]
    report["avg_equal_error_rate"] = float(eer.mean())
    for label, i in label2target.items():
        report[label]["equal_error_rate"] = eer[i]
    report["confusion_matrix"] = sklearn.metrics.confusion_matrix(true_sparse, pred_sparse)
    return report
