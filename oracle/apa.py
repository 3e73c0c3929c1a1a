"""Oracle: Accompaniment Prompt Adherence from three Frechet distances.

Follows src/audio_metrics/metrics/apa.py:5-32.  TEST INFRASTRUCTURE ONLY.
"""
from .fad import frechet_distance


def apa_from_distances(d_y_x, d_y_xp, d_x_xp):
    """apa.py:22-32: clamp the three distances at 0; numerator d(y,x')-d(y,x);
    denominator max(d(x,x'), |numerator|); 0 if the denominator is <= 0;
    otherwise 1/2 + numerator / (2*denominator)."""
    d_y_x = max(0, d_y_x)
    d_y_xp = max(0, d_y_xp)
    d_x_xp = max(0, d_x_xp)
    num = d_y_xp - d_y_x
    den = max(d_x_xp, abs(num))
    if den <= 0:
        return 0.0
    return 0.5 + num / (2 * den)


def apa(candidate, reference, anti_reference, d_x_xp=None):
    """apa.py:9-19."""
    d_y_x = frechet_distance(candidate, reference)
    d_y_xp = frechet_distance(candidate, anti_reference)
    if d_x_xp is None:
        d_x_xp = frechet_distance(reference, anti_reference)
    return apa_from_distances(d_y_x, d_y_xp, d_x_xp)
