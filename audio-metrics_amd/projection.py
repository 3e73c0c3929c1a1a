"""PCA projection of the embeddings (``n_pca``), device-resident.

Mirror of the reference's ``IncrementalPCA`` (src/audio_metrics/projection.py:6-46, a thin
subclass of scikit-learn's): ``partial_fit`` / ``transform`` / ``__getstate__`` /
``__setstate__`` with the same fitted attributes.  scikit-learn takes the SVD of the centred
batch (first call) or of the stacked matrix [diag(s) C ; X - batch_mean ; correction row]
(later calls) and flips signs so that the largest-magnitude entry of every component is
positive.  Here the right singular vectors come from the symmetric eigendecomposition of the
D x D Gram matrix of that same stacked matrix:

    G = C^T diag(s^2) C + sum (x - batch_mean)(x - batch_mean)^T + corr corr^T

with the scatter term from the HIP stats kernel (f64 accumulation); singular values are the
square roots of the eigenvalues.  The D x D eigendecomposition (am_eigh_sym_f64: one-sided Jacobi
in f64) and the N x D by D x p projection (am_project_f64: f64 matrix cores) are this library's own
kernels too; what is left to torch here is O(D^2) glue on device tensors (the rank-one correction,
the sign convention, scalings)."""
import numpy as np
import torch

from . import hip_ops as ops
from .data import ensure_tensor

_ARRAYS = ("components_", "mean_", "var_", "singular_values_", "explained_variance_", "explained_variance_ratio_")


class IncrementalPCA:
    def __init__(self, n_components=None, device=None):
        self.n_components = n_components
        self._device = torch.device(device) if device is not None else None

    # ------------------------------------------------------------ fitting
    def _to_device(self, x):
        x = ensure_tensor(x)
        if not x.is_cuda:
            if self._device is None:
                from .data import default_device
                self._device = default_device()
            x = x.to(self._device)
        elif self._device is None:
            self._device = x.device
        return ops.as_rows(x)                # float64 rows (a float64 embedder) are fitted and projected in f64, as scikit-learn does

    def partial_fit(self, X, y=None, check_input=True, batch_stats=None):
        """scikit-learn's incremental update from one batch.  batch_stats = (n, mean f64[D], cov f64[D, D]) of the batch when
        the caller already holds them - the row-sharded form (one process per GPU): the batch is the union of every rank's
        rows, its statistics come from two all-reduces (distributed.global_stats) and the fit below, a function of those
        statistics and the previous state only, runs replicated on every rank; X is then this rank's shard (possibly
        empty) and is used for its width only."""
        X = self._to_device(X)
        n_samples, n_features = X.shape
        if batch_stats is not None:
            n_samples = int(batch_stats[0])
        first_pass = getattr(self, "components_", None) is None
        if self.n_components is None:
            self.n_components_ = min(n_samples, n_features) if first_pass else self.components_.shape[0]
        elif not self.n_components <= n_features:
            raise ValueError("n_components=%r invalid for n_features=%d, need more rows than columns for "
                             "IncrementalPCA processing" % (self.n_components, n_features))
        elif self.n_components > n_samples and first_pass:
            raise ValueError(f"n_components={self.n_components} must be less or equal to the batch number of samples "
                             f"{n_samples} for the first partial_fit call.")
        else:
            self.n_components_ = self.n_components

        if batch_stats is not None:
            mean_b, cov_b = (ensure_tensor(t).to(X.device, torch.float64) for t in batch_stats[1:])
        else:
            mean_b, cov_b = ops.stats(X)                      # f64 on the device
        if n_samples > 1:
            scatter_b = cov_b * float(n_samples - 1)
        else:
            scatter_b = torch.zeros_like(cov_b)
        var_b = torch.diagonal(scatter_b) / float(n_samples)  # population variance, as _incremental_mean_and_var
        n_seen = int(getattr(self, "n_samples_seen_", 0))
        if n_seen == 0:
            n_total = n_samples
            col_mean, col_var = mean_b, var_b
            gram = scatter_b
            rows = n_samples
        else:
            n_total = n_seen + n_samples
            mean_prev = self.mean_.to(X.device)
            col_mean = (n_seen * mean_prev + n_samples * mean_b) / n_total
            delta = mean_prev - mean_b
            col_var = (self.var_.to(X.device) * n_seen + var_b * n_samples
                       + delta * delta * (n_seen * n_samples / n_total)) / n_total
            corr = np.sqrt((n_seen / n_total) * n_samples) * delta
            c = self.components_.to(X.device)
            s2 = self.singular_values_.to(X.device) ** 2
            gram = (c.T * s2) @ c + scatter_b + torch.outer(corr, corr)
            rows = c.shape[0] + n_samples + 1
        gram = 0.5 * (gram + gram.T)
        evals, vt = ops.eigh_descending(gram)                 # descending; rows of vt = right singular vectors
        evals = evals.clamp_min(0.0)
        # svd_flip(u_based_decision=False): the entry of largest magnitude in each row becomes positive
        idx = vt.abs().argmax(dim=1)
        signs = torch.sign(vt[torch.arange(vt.shape[0], device=vt.device), idx])
        signs[signs == 0] = 1.0
        vt = vt * signs[:, None]
        n_sv = min(rows, n_features)                          # singular values scikit-learn's thin SVD returns
        s = evals.sqrt()[:n_sv]
        explained_variance = s ** 2 / (n_total - 1)
        explained_variance_ratio = s ** 2 / torch.sum(col_var * n_total)

        p = self.n_components_
        self.n_samples_seen_ = n_total
        self.components_ = vt[:p].contiguous()
        self.singular_values_ = s[:p].contiguous()
        self.mean_ = col_mean
        self.var_ = col_var
        self.explained_variance_ = explained_variance[:p].contiguous()
        self.explained_variance_ratio_ = explained_variance_ratio[:p].contiguous()
        if p not in (n_samples, n_features) and p < n_sv:
            self.noise_variance_ = float(explained_variance[p:].mean())
        else:
            self.noise_variance_ = 0.0
        return self

    def fit(self, X, y=None):
        for attr in _ARRAYS + ("n_samples_seen_", "noise_variance_", "n_components_"):
            if hasattr(self, attr):
                delattr(self, attr)
        return self.partial_fit(X)

    # ------------------------------------------------------------ projection
    def transform(self, x):
        """(x - mean_) @ components_^T in f64 -> device tensor [n, n_components]."""
        x = self._to_device(x)
        return ops.project(x, self.mean_.to(x.device), self.components_.to(x.device))

    # ------------------------------------------------------------ state (reference projection.py:23-46)
    def __getstate__(self):
        state = {"n_components": self.n_components}
        for k in _ARRAYS:
            if getattr(self, k, None) is not None:
                state[k] = getattr(self, k).detach().cpu()
        for k in ("n_samples_seen_", "n_components_"):
            if hasattr(self, k):
                state[k] = int(getattr(self, k))
        if hasattr(self, "noise_variance_"):
            state["noise_variance_"] = float(self.noise_variance_)
        return state

    def __setstate__(self, state):
        dev = getattr(self, "_device", None)
        for k, v in state.items():
            if k in _ARRAYS:
                v = ensure_tensor(v).to(torch.float64)
                if dev is not None:
                    v = v.to(dev)
            self.__dict__[k] = v
        self.__dict__.setdefault("_device", dev)
