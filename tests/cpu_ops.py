"""CPU stand-in for ``audio_metrics_amd.hip_ops`` built on the oracle.

TEST INFRASTRUCTURE: lets the world_size-2 gloo tests exercise the sharding and
collective logic of ``audio_metrics_amd.distributed`` without a GPU.  Never used
by the product path."""
import numpy as np
import torch

import oracle
from oracle import kd as okd


def colsum(local):
    return local.double().sum(0)


def scatter(local, mean):
    xc = local.double() - mean
    return xc.T @ xc


def frechet(mu_x, cov_x, mu_y, cov_y, *a):
    return {"fd": oracle.frechet_from_stats(mu_x, cov_x, mu_y, cov_y)}


def kd_poly(x, y, idx1, idx2, gamma, coef0, degree):
    x, y = x.numpy(), y.numpy()
    out = np.zeros(len(idx1))
    for s in range(len(idx1)):
        a, b = x[idx1[s].numpy()], y[idx2[s].numpy()]
        out[s] = okd.mmd2_unbiased(okd.poly_kernel(a, a, degree, gamma, coef0), okd.poly_kernel(a, b, degree, gamma, coef0),
                                   okd.poly_kernel(b, b, degree, gamma, coef0))
    return torch.as_tensor(out)


def knn_radii(x, k, columns=None):
    y = x if columns is None else columns
    return torch.kthvalue(torch.cdist(x, y), k=k + 1, dim=-1)[0]


def prdc_counts(ref, cand, r_ref, r_cand):
    d = torch.cdist(ref, cand)
    return ((d < r_ref[:, None]).sum(0).to(torch.int32), (d < r_cand[None, :]).any(1).to(torch.uint8),
            (d.min(1)[0] < r_ref).to(torch.uint8))


def prdc_reduce(col, rany, rcov):
    return torch.tensor([int((col > 0).sum()), int(rany.sum()), int(col.sum()), int(rcov.sum())], dtype=torch.int64)
