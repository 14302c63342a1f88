"""
Counterpart of lidbox/metrics.py: AverageDetectionCost / SparseAverageDetectionCost (reference
metrics.py:6-119), C_avg of Li, Ma & Lee (2013) eq. 32 as a streaming metric.  The reference
materialises four [B, N, Th] tensors per update; the HIP kernel counts straight into the
[N, Th] / [N, N, Th] state.
"""
import torch

from . import _native as nv


class AverageDetectionCost:
    def __init__(self, N, thresholds, C_miss=1.0, C_fa=1.0, P_tar=0.5, name="C_avg", device=None):
        if N < 2:
            raise ValueError("C_avg is undefined for less than 2 classes.")                  # :20
        th = torch.as_tensor(thresholds, dtype=torch.float32)
        if th.dim() != 1:
            raise ValueError("Thresholds must be an array of decision scores.")              # :21
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.N, self.name = int(N), name
        self.thresholds = th.to(self.device).contiguous()
        self.C_miss, self.C_fa, self.P_tar = float(C_miss), float(C_fa), float(P_tar)
        Th = th.numel()
        f32 = dict(dtype=torch.float32, device=self.device)
        self.fn = torch.zeros((N, Th), **f32)
        self.tp = torch.zeros((N, Th), **f32)
        self.fp_pairs = torch.zeros((N, N, Th), **f32)
        self.tn_pairs = torch.zeros((N, N, Th), **f32)

    def reset_states(self):
        """reference metrics.py:46-48"""
        for var in (self.fn, self.tp, self.fp_pairs, self.tn_pairs):
            var.zero_()

    def _update_sparse(self, labels, predictions):
        s = nv.require_gpu_tensor(predictions, "predictions").to(torch.float32).contiguous()
        if s.dim() != 2 or s.shape[1] != self.N:
            raise ValueError("predictions must be [batch_size, %d]" % self.N)
        y = labels.to(device=s.device, dtype=torch.int32).reshape(-1).contiguous()
        with torch.cuda.device(s.device):
            nv.check(nv.lib.lidbox_cavg_update(nv.ptr(s), nv.ptr(y), s.shape[0], self.N, nv.ptr(self.thresholds),
                                               self.thresholds.numel(), nv.ptr(self.tp), nv.ptr(self.fn),
                                               nv.ptr(self.fp_pairs), nv.ptr(self.tn_pairs), nv.current_stream()))

    def update_state(self, true_positives, predictions, **kwargs):
        """reference metrics.py:51-71: one-hot float labels [B, N], scores [B, N]."""
        tp = torch.as_tensor(true_positives)
        self._update_sparse(torch.argmax(tp, dim=-1), predictions)                           # :56

    def result(self, return_per_threshold=False):
        """reference metrics.py:73-103: min over thresholds of C_avg."""
        Th = self.thresholds.numel()
        out = torch.empty(1, dtype=torch.float32, device=self.device)
        c = torch.empty(Th, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            nv.check(nv.lib.lidbox_cavg_result(nv.ptr(self.tp), nv.ptr(self.fn), nv.ptr(self.fp_pairs),
                                               nv.ptr(self.tn_pairs), self.N, Th, self.C_miss, self.C_fa, self.P_tar,
                                               nv.ptr(c), nv.ptr(out), nv.current_stream()))
        return (out[0], c) if return_per_threshold else out[0]

    def counters(self):
        """the four state tensors"""
        return [self.tp, self.fn, self.fp_pairs, self.tn_pairs]

    def sync_counters(self, group=None):
        """Data parallelism: every rank counts its own shard; one all-reduce(sum) of the counters
        ([N,Th] x 2 + [N,N,Th] x 2, 8 MB at N = Th = 100) before `result()` gives the global metric.  Counts are
        integers held in float32, so the sum is exact and independent of the reduction order.  No-op without an
        initialised process group.  Call it once per evaluation, then `reset_states()`."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            return self
        for t in self.counters():
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return self


class SparseAverageDetectionCost(AverageDetectionCost):
    """reference metrics.py:114-119"""

    def update_state(self, true_positives, predictions, **kwargs):
        self._update_sparse(torch.as_tensor(true_positives), predictions)
