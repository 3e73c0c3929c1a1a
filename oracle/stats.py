"""Oracle: per-set sufficient statistics (n, mean, cov) with Chan merging.

Follows the reference's ``AudioMetricsData`` (src/audio_metrics/data.py:18-112).
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
import numpy as np
import torch


def _as_tensor(x):
    # data.py:6-9 (ensure_tensor)
    return x if isinstance(x, torch.Tensor) else torch.as_tensor(x)


def batch_stats(emb):
    """Mean / unbiased covariance of one batch, computed in the batch's own
    dtype and only then widened to f64 (data.py:38-44).  A single row has a
    zero covariance (data.py:40-42)."""
    emb = _as_tensor(emb)
    n = emb.shape[0]
    mean = emb.mean(dim=0).to(torch.float64)
    if n == 1:
        d = emb.shape[-1]
        cov = torch.zeros((d, d), dtype=torch.float64)
    else:
        cov = torch.cov(emb.T).to(torch.float64)
    return n, mean, cov


def chan_merge(n1, mean1, cov1, n2, mean2, cov2):
    """Pairwise (Chan) merge of two (n, mean, cov) triples in f64
    (data.py:77-94): weights (n1-1)/(n-1), (n2-1)/(n-1) and
    (n1*n2/n)/(n-1) on the outer product of the mean difference."""
    n = n1 + n2
    mean = (n1 * mean1 + n2 * mean2) / n
    delta = mean1 - mean2
    outer = delta[:, None] * delta[None, :]
    w1 = (n1 - 1) / (n - 1)
    w2 = (n2 - 1) / (n - 1)
    wd = (n1 * n2 / n) / (n - 1)
    cov = w1 * cov1 + w2 * cov2 + wd * outer
    return n, mean, cov


class OracleData:
    """Minimal stand-in for ``AudioMetricsData`` (data.py:18-112): running
    stats, optional concatenated embedding store, cached radii."""

    def __init__(self, store_embeddings=True):
        self.n = None
        self.mean = None
        self.cov = None
        self.store_embeddings = store_embeddings
        self.embeddings = None
        self.radii = {}

    def __len__(self):
        return self.n or 0                      # data.py:74-75

    def add(self, emb):                         # data.py:37-47
        emb = _as_tensor(emb)
        n, mean, cov = batch_stats(emb)
        self._merge(n, mean, cov)
        if self.store_embeddings:
            self._append(emb)
        return self

    def recompute_stats(self):                  # data.py:49-58
        if self.embeddings is None:
            return
        e = self.embeddings
        self.n = len(e)
        self.mean = e.mean(dim=0).to(torch.float64)
        if self.n == 1:
            self.cov = torch.zeros((1, 1), dtype=torch.float64)   # reference quirk, data.py:56
        else:
            self.cov = torch.cov(e.T).to(torch.float64)

    def get_radii(self, k):                     # data.py:60-66
        from .prdc import knn_radii
        key = f"radii_{k}"
        r = self.radii.get(key)
        if r is None and self.embeddings is not None:
            r = knn_radii(self.embeddings, k)
            self.radii[key] = r
        return r

    def merge(self, other):                     # data.py:96-106 (__iadd__)
        if other.n is None:
            return self
        if self.n is None:
            self.store_embeddings = other.store_embeddings
        assert self.store_embeddings == other.store_embeddings
        self._merge(other.n, other.mean, other.cov)
        if self.store_embeddings:
            self._append(other.embeddings)
        return self

    def _merge(self, n, mean, cov):
        if self.n is None:
            self.n, self.mean, self.cov = n, mean, cov
        else:
            self.n, self.mean, self.cov = chan_merge(self.n, self.mean, self.cov, n, mean, cov)

    def _append(self, emb):                     # data.py:68-72
        if self.embeddings is None:
            self.embeddings = emb.clone()
        else:
            self.embeddings = torch.cat((self.embeddings, emb))
