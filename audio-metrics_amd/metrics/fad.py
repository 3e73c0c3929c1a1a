"""Frechet distance on the device (reference src/audio_metrics/metrics/fad.py:8-31)."""
import torch

from .. import hip_ops as ops
from ..data import AudioMetricsData, ensure_tensor

NS_MAX_ITER = 64
NS_TOL = 1e-13

last_info = {}     # iterations / residual of the most recent call (diagnostics)


def frechet_distance(x: AudioMetricsData, y: AudioMetricsData, device=None):
    return _frechet_distance(x.mean, x.cov, y.mean, y.cov, device=device)


def _frechet_distance(mu_x, sigma_x, mu_y, sigma_y, device=None) -> float:
    """|mu_x-mu_y|^2 + tr(sigma_x) + tr(sigma_y) - 2 tr sqrt(sigma_x sigma_y), f64.
    ``device`` selects the MI355X to run on (the reference's optional kwarg,
    fad.py:24-27); host tensors are uploaded."""
    if device is None:
        device = next((t.device for t in (mu_x, sigma_x, mu_y, sigma_y)
                       if isinstance(t, torch.Tensor) and t.is_cuda), None)
    if device is None:
        from ..data import default_device
        device = default_device()
    mu_x, sigma_x, mu_y, sigma_y = (ensure_tensor(t).to(device, torch.float64) for t in (mu_x, sigma_x, mu_y, sigma_y))
    res = ops.frechet(mu_x, sigma_x, mu_y, sigma_y, NS_MAX_ITER, NS_TOL)
    last_info.clear()
    last_info.update(res)
    return res["fd"]
