"""Precision / recall / density / coverage on the device
(reference src/audio_metrics/metrics/prdc.py:4-50)."""
from .. import hip_ops as ops


def nearest_neighbour_distances(input_features, nearest_k):
    """Distance to the k-th nearest neighbour of every row (the (k+1)-th smallest
    entry of its self-distance row) - prdc.py:4-14."""
    return ops.knn_radii(input_features, nearest_k)


def prdc(reference, candidate, nearest_k):
    """dict(precision, recall, density, coverage); arguments are
    ``AudioMetricsData``-like objects exposing ``embeddings`` and ``get_radii``."""
    radii = [side.get_radii(nearest_k) for side in (reference, candidate)]      # cached on the objects (data.py:60-66)
    prepared = [side.prepared() if hasattr(side, "prepared") else None for side in (reference, candidate)]
    col_count, row_any, row_cover = ops.prdc_counts(reference.embeddings, candidate.embeddings, radii[0], radii[1],
                                                    prepared_ref=prepared[0], prepared_cand=prepared[1])
    n_prec, n_rec, sum_cnt, n_cov = (int(v) for v in ops.prdc_reduce(col_count, row_any, row_cover).cpu().tolist())
    n_ref, n_cand = row_any.numel(), col_count.numel()
    # means of 0/1 (and integer) values in f64, as the reference's .double().mean()
    precision = n_prec / n_cand
    recall = n_rec / n_ref
    density = (1.0 / float(nearest_k)) * (sum_cnt / n_cand)
    coverage = n_cov / n_ref
    return dict(precision=precision, recall=recall, density=density, coverage=coverage)
