"""
Counterpart of lidbox/losses.py: SparseAngularProximity (reference losses.py:4-52), the
angular-proximity loss of Gelly & Gauvain (2017).  Reference directions are the first N one-hot
axes of R^D (:20), so theta = acos(z[:, :N]) (:42-49).
"""
import torch

from . import _native as nv


class SparseAngularProximity:
    def __init__(self, N, D, delta_weight=1.0, name="AP"):
        if N < 1:
            raise ValueError("Must have at least 1 class")                                   # :14
        if D < N:
            raise ValueError("Language vector dimension cannot be less than number of classes")   # :15
        if not delta_weight > 0:
            raise ValueError("Non-positive delta weight would cause correct classifications to have larger "
                             "loss values than incorrect classifications.")                  # :16
        self.N, self.D, self.delta_weight, self.name = int(N), int(D), float(delta_weight), name

    def _run(self, y_true_sparse, y_pred, want_grad, scale):
        z = nv.require_gpu_tensor(y_pred, "y_pred", torch.float32).contiguous()
        if z.dim() != 2 or z.shape[1] != self.D:
            raise ValueError("y_pred must be [batch_size, %d]" % self.D)
        y = torch.as_tensor(y_true_sparse, device=z.device).to(torch.int32).reshape(-1).contiguous()
        B = z.shape[0]
        loss = torch.empty(B, dtype=torch.float32, device=z.device)
        dz = torch.empty_like(z) if want_grad else None
        with torch.cuda.device(z.device):
            nv.check(nv.lib.lidbox_ap_loss_fwd_bwd(nv.ptr(z), nv.ptr(y), B, self.D, self.N, self.delta_weight,
                                                   float(scale), nv.ptr(loss), nv.ptr(dz), nv.current_stream()))
        return loss, dz

    def call(self, y_true_sparse, y_pred):
        """reference losses.py:25-40: per-example loss [batch_size]."""
        return self._run(y_true_sparse, y_pred, False, 1.0)[0]

    def __call__(self, y_true_sparse, y_pred):
        """Keras Loss.__call__: mean over the batch (SUM_OVER_BATCH_SIZE)."""
        return self._mean(self.call(y_true_sparse, y_pred))

    def loss_and_grad(self, y_true_sparse, y_pred, scale=None):
        """(mean loss, d mean loss / d y_pred); scale defaults to 1/batch."""
        B = y_pred.shape[0]
        loss, dz = self._run(y_true_sparse, y_pred, True, (1.0 / B) if scale is None else scale)
        return self._mean(loss), dz

    @staticmethod
    def _mean(per_example):
        """Keras' SUM_OVER_BATCH_SIZE reduction: lidbox_mean (one workgroup, fixed order)"""
        out = torch.zeros((), dtype=torch.float32, device=per_example.device)
        if per_example.numel():
            with torch.cuda.device(per_example.device):
                nv.check(nv.lib.lidbox_mean(nv.ptr(per_example), per_example.numel(), nv.ptr(out), nv.current_stream()))
        return out

    def theta(self, z):
        """reference losses.py:42-49 (elementwise acos of the first N coordinates): -predict, on the same kernel."""
        out = self.predict(z)
        if out.numel():
            with torch.cuda.device(out.device):
                nv.check(nv.lib.lidbox_scale(nv.ptr(out), out.numel(), -1.0, nv.current_stream()))
        return out

    def predict(self, z):
        """reference losses.py:51-52: -acos(z[:, :N]) (lidbox_neg_acos, the kernel the scoring path uses)"""
        z = nv.require_gpu_tensor(z, "z", torch.float32).contiguous()
        B, D = z.shape
        out = torch.empty((B, self.N), dtype=torch.float32, device=z.device)
        if out.numel():
            with torch.cuda.device(z.device):
                nv.check(nv.lib.lidbox_neg_acos(nv.ptr(z), B, D, self.N, nv.ptr(out), nv.current_stream()))
        return out
