"""Classification and regression metrics on the reduced MC output, accumulated on the device.

Mirror of reference src/metrics.py `ClassificationMetric` (:355-430) and its members Error (:8-33), NLL (:36-62),
Brier (:65-91), PredictiveEntropy (:94-116) and the 10-bin L1 calibration error (:381-383).  The reference feeds these
from `_evaluate_with_loader` (experiments/utils.py:357) right after the MC reduction.
"""
import torch

from . import _lib

SLOTS = 34


class ClassificationMetric:
    metric_labels = ["nll", "error", "entropy", "brier", "ece"]

    def __init__(self, output_size, writer=None):
        self.output_size = int(output_size)
        self.sums, self.count = None, 0

    @torch.no_grad()
    def update(self, output, target, **kwargs):
        """output: [B, C] predictive mean on the GPU; target: [B] integer labels."""
        if output.device.type != "cuda":
            raise RuntimeError("qbnn metrics run on an MI355X only (no CPU fallback)")
        output = output.detach().to(torch.float32).contiguous()
        target = target.to(device=output.device, dtype=torch.int64).contiguous()
        B, Cc = output.shape
        nb = (B + 255) // 256
        partial = torch.empty((nb, SLOTS), dtype=torch.float32, device=output.device)
        _lib.check(_lib.lib().qbnn_classification_metrics(_lib.ptr(output), _lib.ptr(target), B, Cc, _lib.ptr(partial), _lib.current_stream()))
        s = partial.to(torch.float64).sum(0)
        self.sums = s if self.sums is None else self.sums + s
        self.count += B

    def _s(self, i):
        return float(self.sums[i])

    @property
    def error(self):
        return self._s(0) / self.count

    @property
    def nll(self):
        return self._s(1) / self.count

    @property
    def brier(self):
        return self._s(2) / self.count

    @property
    def entropy(self):
        return self._s(3) / self.count

    @property
    def ece(self):
        s = self.sums.cpu()
        n, conf, acc = s[4:14], s[14:24], s[24:34]
        m = n > 0
        return float(((acc[m] / n[m] - conf[m] / n[m]).abs() * (n[m] / self.count)).sum())

    def get_key_metric(self):
        return self.error

    def compute(self):
        return {k: getattr(self, k) for k in self.metric_labels}


class RegressionMetric:
    """reference src/metrics.py `RegressionMetric` (:433-504) with its members RegressionNegativeLogLikelihood (:119-161),
    MeanSquaredError (:164-191), RootMeanSquaredError (:194-199), MeanAbsoluteError (:202-229): fed with the MC-reduced
    `(mean, variance)` pair of `mc_predict_regression` (experiments/utils.py:348-353, :357)."""
    metric_labels = ["nll", "rmse", "mse", "mae"]

    def __init__(self, output_size=1, writer=None):
        self.sums, self.count = None, 0

    @torch.no_grad()
    def update(self, output, target, **kwargs):
        """output: (mean [B, 1], var [B, 1]) on the GPU (var may be None: unit variance); target: [B] or [B, 1]."""
        mean, var = output[0], output[1]
        if mean.device.type != "cuda":
            raise RuntimeError("qbnn metrics run on an MI355X only (no CPU fallback)")
        mean = mean.detach().to(torch.float32).reshape(-1).contiguous()
        var = None if var is None else var.detach().to(torch.float32).reshape(-1).contiguous()
        target = target.to(device=mean.device, dtype=torch.float32).reshape(-1).contiguous()
        B = mean.numel()
        if target.numel() != B:
            raise ValueError("target and prediction sizes differ")
        nb = (B + 255) // 256
        partial = torch.empty((nb, 3), dtype=torch.float32, device=mean.device)
        _lib.check(_lib.lib().qbnn_regression_metrics(_lib.ptr(mean), _lib.ptr(var), _lib.ptr(target), B, _lib.ptr(partial), _lib.current_stream()))
        s = partial.to(torch.float64).sum(0)
        self.sums = s if self.sums is None else self.sums + s
        self.count += B

    @property
    def nll(self):
        return float(self.sums[0]) / self.count

    @property
    def mse(self):
        return float(self.sums[1]) / self.count

    @property
    def rmse(self):
        return self.mse ** 0.5

    @property
    def mae(self):
        return float(self.sums[2]) / self.count

    def get_key_metric(self):
        return self.rmse

    def compute(self):
        return {k: getattr(self, k) for k in self.metric_labels}
