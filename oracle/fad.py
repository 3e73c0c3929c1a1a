"""Oracle: Frechet distance between two Gaussians given (mean, cov).

Follows src/audio_metrics/metrics/fad.py:8-31.  TEST INFRASTRUCTURE ONLY.
"""
import torch


def frechet_from_stats(mu_x, cov_x, mu_y, cov_y):
    """||mu_x-mu_y||^2 + tr(cov_x) + tr(cov_y) - 2*sum(Re sqrt(eig(cov_x cov_y)))
    with general (complex) eigenvalues of the non-symmetric product, all in the
    inputs' dtype (f64 on the hot path) - fad.py:28-31."""
    mu_x, cov_x, mu_y, cov_y = (torch.as_tensor(t) for t in (mu_x, cov_x, mu_y, cov_y))
    a = (mu_x - mu_y).square().sum(dim=-1)
    b = cov_x.trace() + cov_y.trace()
    c = torch.linalg.eigvals(cov_x @ cov_y).sqrt().real.sum(dim=-1)
    return (a + b - 2 * c).item()


def trace_sqrt_product(cov_x, cov_y):
    """The ``c`` term alone (fad.py:30); used to check the device Newton-Schulz."""
    cov_x, cov_y = torch.as_tensor(cov_x), torch.as_tensor(cov_y)
    return torch.linalg.eigvals(cov_x @ cov_y).sqrt().real.sum(dim=-1).item()


def frechet_distance(x, y):
    """fad.py:8-13 on objects exposing ``.mean`` / ``.cov``."""
    return frechet_from_stats(x.mean, x.cov, y.mean, y.cov)
