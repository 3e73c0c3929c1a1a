"""Oracle: k-NN radii and precision / recall / density / coverage.

Follows src/audio_metrics/metrics/prdc.py:4-50.  TEST INFRASTRUCTURE ONLY.
"""
import torch


def knn_radii(features, k):
    """(k+1)-th smallest entry of each row of the self-distance matrix, i.e. the
    distance to the k-th neighbour once the (approximately zero) self distance
    is counted (prdc.py:12-13).  ``torch.cdist`` default compute mode."""
    features = torch.as_tensor(features)
    d = torch.cdist(features, features)
    return torch.kthvalue(d, k=k + 1, dim=-1)[0]


def prdc_from_features(ref, cand, r_ref, r_cand, k, return_counts=False):
    """prdc.py:34-50 given precomputed radii.  Strict ``<``; means in f64."""
    ref, cand = torch.as_tensor(ref), torch.as_tensor(cand)
    d = torch.cdist(ref, cand)                                   # [Nr, Nc]
    inside_ref = d < r_ref[:, None]
    precision = inside_ref.any(dim=0).double().mean().item()
    recall = (d < r_cand[None, :]).any(dim=1).double().mean().item()
    col_count = inside_ref.sum(dim=0)
    density = (1.0 / float(k)) * col_count.double().mean().item()
    row_min = d.min(dim=1)[0]
    coverage = (row_min < r_ref).double().mean().item()
    out = dict(precision=precision, recall=recall, density=density, coverage=coverage)
    if return_counts:
        return out, dict(col_count=col_count, row_any=(d < r_cand[None, :]).any(dim=1), row_min=row_min)
    return out


def prdc(reference, candidate, k):
    """prdc.py:18-50 on objects exposing ``.embeddings`` / ``.get_radii``."""
    r_ref = reference.get_radii(k)
    r_cand = candidate.get_radii(k)
    return prdc_from_features(reference.embeddings, candidate.embeddings, r_ref, r_cand, k)


def prdc_blocked(ref, cand, k, block=4096, return_counts=False):
    """Same quantities without materialising N x N matrices (row blocks of the
    same torch calls).  Used as the timed CPU baseline and as the generator of
    ``tests/golden/bench_prdc.npz`` at sizes where the reference's own N x N
    formulation does not fit host memory; values can differ from ``prdc`` in the
    last f32 bit of individual distances (``torch.cdist`` picks its summation
    blocking by shape).  ``tests/test_oracle_golden.py::test_prdc_blocked_vs_reference_large``
    pins it against the REFERENCE's own outputs at 33 000 - 40 000 rows
    (radii, integer column counts, row flags)."""
    ref, cand = torch.as_tensor(ref), torch.as_tensor(cand)

    def radii(x):
        out = torch.empty(len(x), dtype=x.dtype)
        for s in range(0, len(x), block):
            out[s:s + block] = torch.kthvalue(torch.cdist(x[s:s + block], x), k=k + 1, dim=-1)[0]
        return out

    r_ref, r_cand = radii(ref), radii(cand)
    col_count = torch.zeros(len(cand), dtype=torch.int64)
    row_any = torch.zeros(len(ref), dtype=torch.bool)
    row_min = torch.empty(len(ref), dtype=ref.dtype)
    for s in range(0, len(ref), block):
        d = torch.cdist(ref[s:s + block], cand)
        col_count += (d < r_ref[s:s + block, None]).sum(dim=0)
        row_any[s:s + block] = (d < r_cand[None, :]).any(dim=1)
        row_min[s:s + block] = d.min(dim=1)[0]
    out = dict(precision=(col_count > 0).double().mean().item(),
               recall=row_any.double().mean().item(),
               density=(1.0 / float(k)) * col_count.double().mean().item(),
               coverage=(row_min < r_ref).double().mean().item())
    if return_counts:
        return out, dict(r_ref=r_ref, r_cand=r_cand, col_count=col_count, row_any=row_any, row_cover=row_min < r_ref)
    return out
