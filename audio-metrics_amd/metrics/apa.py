"""Accompaniment Prompt Adherence (reference src/audio_metrics/metrics/apa.py:5-32):
three device Frechet distances and the scalar combination (am_apa_f64)."""
from .. import hip_ops as ops
from ..data import AudioMetricsData
from .fad import frechet_distance


def apa_compute_d_x_xp(reference: AudioMetricsData, anti_reference: AudioMetricsData):
    return frechet_distance(reference, anti_reference)


def apa(candidate: AudioMetricsData, reference: AudioMetricsData, anti_reference: AudioMetricsData,
        d_x_xp=None):
    d_y_x = frechet_distance(candidate, reference)
    d_y_xp = frechet_distance(candidate, anti_reference)
    if d_x_xp is None:
        d_x_xp = frechet_distance(reference, anti_reference)
    return _apa(d_y_x, d_y_xp, d_x_xp)


def _apa(d_y_x, d_y_xp, d_x_xp):
    return ops.apa_scalar(d_y_x, d_y_xp, d_x_xp)
