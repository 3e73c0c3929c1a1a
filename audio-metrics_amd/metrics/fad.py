"""Frechet distance on the device (reference src/audio_metrics/metrics/fad.py:8-31)."""
import torch

from .. import hip_ops as ops
from ..data import AudioMetricsData, ensure_tensor

NS_MAX_ITER = 64
NS_TOL = 1e-13

last_info = {}     # iterations / residual of the most recent call (diagnostics)

RANK_DEFICIENT_NOTE = (
    "Frechet distance of a set with no more rows than dimensions ({n} rows, {d} dimensions) from float32 statistics: the "
    "reference's own value here contains the square roots of the rounding dust its float32 torch.cov leaves in the null space "
    "(fad.py:30) - it moves by up to 1.5e-4 relative with the batch size of its add() calls and by 2.6e-4 ... 3.6e-4 between "
    "float32 and float64 rows.  This build returns the dust-free value (= the reference fed float64 rows, to 1e-7); expect up "
    "to 4e-4 relative deviation from what the reference prints for float32 rows.")


def warn_if_rank_deficient(n_x, n_y, d, rows_dtype=None):
    """One RuntimeWarning per call where the reference's float32 result is not defined to 1e-4 (not silently).
    rows_dtype = dtype of the rows the statistics came from, when known: float64 rows carry no such dust."""
    import warnings
    if rows_dtype == torch.float64 or n_x is None or n_y is None:
        return
    n = min(int(n_x), int(n_y))
    if 1 < n <= int(d):
        warnings.warn(RANK_DEFICIENT_NOTE.format(n=n, d=int(d)), RuntimeWarning, stacklevel=3)


def frechet_distance(x: AudioMetricsData, y: AudioMetricsData, device=None):
    dtypes = {getattr(x, "stats_rows_dtype", None), getattr(y, "stats_rows_dtype", None)}
    warn_if_rank_deficient(getattr(x, "n", None), getattr(y, "n", None), ensure_tensor(x.mean).numel(),
                           torch.float64 if dtypes == {torch.float64} else None)
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
