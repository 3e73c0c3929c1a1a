"""Context + stem -> mono mix (host-side DSP upstream of the embedder; protocol
``f(audio[n, 2], sr=int) -> audio[n]`` as in the reference's mix_functions.py:335-344).

The peak-based mixers are restated here in numpy.  The loudness-based ones (L0/L1/L2,
BS.1770 gating + limiter) depend on pyloudnorm / numpy_audio_limiter, which are outside
this build's scope; they are resolved lazily and raise a clear error when unavailable."""
from functools import partial

import numpy as np


def mix_tracks_peak_preserve(audio, sr):
    """Average the channels, then rescale to the peak of the original waveforms
    (reference mix_functions.py:209-227)."""
    assert audio.ndim == 2
    if audio.shape[1] == 1:
        return audio[:, 0]
    peak = np.abs(audio).max()
    if peak <= 1e-5:
        return audio[:, 0]
    mix = audio.mean(axis=1)
    mix *= peak / np.abs(mix).max()
    return mix


def mix_tracks_peak_normalize(audio, sr, stem_db_red=0.0, out_db=0.0):
    """Peak-normalise each channel (the stem `stem_db_red` dB below the context), sum,
    and peak-normalise the mix to `out_db` dBFS (reference mix_functions.py:230-250)."""
    assert audio.ndim == 2
    out_gain = np.power(10.0, out_db / 20.0)
    stem_gain = np.power(10.0, stem_db_red / 20.0)
    if audio.shape[1] == 1:
        mix = audio[:, 0]
    else:
        peaks = np.abs(audio).max(axis=0, keepdims=True)
        peaks[0, 1] *= stem_gain
        mix = (audio / peaks).sum(axis=1)
    mix *= out_gain / np.abs(mix).max()
    return mix


def _loudness_mixer(stem_db_red, out_db):
    def mix(audio, sr):
        try:
            import pyloudnorm  # noqa: F401
        except ImportError as e:
            raise ImportError("the loudness-based mix functions (L0/L1/L2) need `pyloudnorm`, which is not installed; "
                              "pass mix_function='P0' (peak based) or your own callable f(audio[n,2], sr)->audio[n]") from e
        raise NotImplementedError("BS.1770 loudness mixing is outside the scope of this build (SURVEY.md section 2 row 14)")
    mix.stem_db_red, mix.out_db = stem_db_red, out_db
    return mix


MIX_FUNCTIONS = dict(
    PP=mix_tracks_peak_preserve,
    P0=partial(mix_tracks_peak_normalize, stem_db_red=-0, out_db=-3),
    P1=partial(mix_tracks_peak_normalize, stem_db_red=-3, out_db=-3),
    P2=partial(mix_tracks_peak_normalize, stem_db_red=-6, out_db=-3),
    L0=_loudness_mixer(0, -20),
    L1=_loudness_mixer(-3, -20),
    L2=_loudness_mixer(-6, -20),
)
DEFAULT_MIX_FUNCTION = "L0"
