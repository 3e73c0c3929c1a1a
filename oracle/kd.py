"""Oracle: Kernel Distance (KID-style unbiased MMD^2, polynomial kernel).

Follows src/audio_metrics/metrics/kd.py:17-26 (constants), 38-83 (mmd2),
112-116 (polynomial kernel), 119-124, 127-194.  TEST INFRASTRUCTURE ONLY.
"""
import numpy as np
import torch

SUBSETS = 100          # kd.py:19
SUBSET_SIZE = 1000     # kd.py:20
DEGREE = 3             # kd.py:22
GAMMA = None           # kd.py:23  -> 1 / D
COEF0 = 1              # kd.py:24
SEED = 1234            # kd.py:176


def _to_numpy(x):
    return x.cpu().numpy() if torch.is_tensor(x) else np.asarray(x)


def effective_subset_size(n1, n2, subset_size=SUBSET_SIZE):
    """kd.py:158-168: shrink to max(1, n_min // 2) when subset_size >= n_min."""
    n_min = min(n1, n2)
    return max(1, n_min // 2) if subset_size >= n_min else subset_size


def draw_subsets(n1, n2, subsets=SUBSETS, subset_size=SUBSET_SIZE, seed=SEED):
    """Index table the reference would draw: one PCG64 generator, per subset a
    draw without replacement from set 1 *then* from set 2 (kd.py:176, 185-186).
    Returns (idx1, idx2) int64 arrays of shape (subsets, m)."""
    m = effective_subset_size(n1, n2, subset_size)
    rng = np.random.default_rng(seed)
    idx1 = np.empty((subsets, m), dtype=np.int64)
    idx2 = np.empty((subsets, m), dtype=np.int64)
    for s in range(subsets):
        idx1[s] = rng.choice(n1, m, replace=False)
        idx2[s] = rng.choice(n2, m, replace=False)
    return idx1, idx2


def poly_kernel(x, y, degree=DEGREE, gamma=GAMMA, coef0=COEF0):
    """(x y^T * gamma + coef0) ** degree in the inputs' dtype (kd.py:112-116)."""
    if gamma is None:
        gamma = 1.0 / x.shape[1]
    return (np.matmul(x, y.T) * gamma + coef0) ** degree


def rbf_kernel(x, y, sigma=10.0):
    """exp(-|x-y|^2 / (2 sigma^2)) with scipy's sqeuclidean cdist (f64) - kd.py:86-109."""
    from scipy.spatial.distance import cdist
    return np.exp(-cdist(x, y, "sqeuclidean") / (2 * sigma ** 2))


def mmd2_unbiased(k_xx, k_xy, k_yy):
    """Unbiased MMD^2 estimate (kd.py:50-79, ``mmd_est='unbiased'``): within-set
    sums drop the diagonal and are divided by m(m-1); the cross term keeps its
    diagonal and is divided by m^2."""
    m = k_xx.shape[0]
    assert k_xx.shape == (m, m) and k_xy.shape == (m, m) and k_yy.shape == (m, m)
    diag_x = np.diagonal(k_xx)
    diag_y = np.diagonal(k_yy)
    kt_xx = (k_xx.sum(axis=1) - diag_x).sum()
    kt_yy = (k_yy.sum(axis=1) - diag_y).sum()
    k_xy_sum = k_xy.sum(axis=0).sum()
    out = (kt_xx + kt_yy) / (m * (m - 1))
    out -= 2 * k_xy_sum / (m * m)
    return out


def kid_from_features(f1, f2, subsets=SUBSETS, subset_size=SUBSET_SIZE, degree=DEGREE,
                      gamma=GAMMA, coef0=COEF0, seed=SEED, return_all=False, kernel_type="polynomial", sigma=10.0):
    """kd.py:127-194 (polynomial kernel by default; kernel_type="rbf" selects kd.py:136-140)."""
    f1, f2 = _to_numpy(f1), _to_numpy(f2)
    assert f1.ndim == 2 and f2.ndim == 2 and f1.shape[1] == f2.shape[1]
    n1, n2 = len(f1), len(f2)
    assert n1 and n2
    idx1, idx2 = draw_subsets(n1, n2, subsets, subset_size, seed)
    mmds = np.zeros(subsets)
    for s in range(subsets):
        a, b = f1[idx1[s]], f2[idx2[s]]
        if kernel_type == "rbf":
            mmds[s] = mmd2_unbiased(rbf_kernel(a, a, sigma), rbf_kernel(a, b, sigma), rbf_kernel(b, b, sigma))
            continue
        mmds[s] = mmd2_unbiased(poly_kernel(a, a, degree, gamma, coef0),
                                poly_kernel(a, b, degree, gamma, coef0),
                                poly_kernel(b, b, degree, gamma, coef0))
    out = {"kernel_distance_mean": float(np.mean(mmds)),
           "kernel_distance_std": float(np.std(mmds))}
    return (out, mmds) if return_all else out


def kernel_distance(x, y):
    """kd.py:29-35 on objects exposing ``.embeddings`` (x is features_1)."""
    return kid_from_features(x.embeddings, y.embeddings)
