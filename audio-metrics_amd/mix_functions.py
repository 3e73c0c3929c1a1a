"""Context + stem -> mono mix (host-side DSP upstream of the embedder; protocol
``f(audio[n, 2], sr=int) -> audio[n]`` as in the reference's mix_functions.py:335-344).

The peak-based mixers are restated here in numpy.  The loudness-based ones (L0/L1/L2,
BS.1770 gating + limiter) depend on pyloudnorm / numpy_audio_limiter, which are outside
this build's scope: ``resolve_mix_function`` rejects them when the object is constructed."""
from functools import partial

import numpy as np


def _peak(x):
    """max |x| of a 1-D array without the |x| temporary (np.abs(x).max() for finite input)."""
    return max(x.max(), -x.min())


def mix_tracks_peak_preserve(audio, sr):
    """Channel average rescaled to the largest peak of the input channels; a single channel, or (near-)silent input,
    passes its first channel through (behaviour of mix_functions.py:209-227).  Written column-wise: the reference's
    np.mean / np.abs over the [n, 2] array walk a 2-element inner axis, 5x slower for the same f32 values."""
    if audio.ndim != 2:
        raise AssertionError("expected audio of shape [n_samples, channels]")
    first = audio[:, 0]
    if audio.shape[1] < 2:
        return first
    if audio.shape[1] != 2:
        loudest = np.abs(audio).max()
        if not loudest > 1e-5:
            return first
        mono = np.mean(audio, axis=1)
        mono *= loudest / np.abs(mono).max()
        return mono
    second = audio[:, 1]
    loudest = max(_peak(first), _peak(second))
    if not loudest > 1e-5:
        return first
    mono = first + second                      # np.mean over the channel pair: (a + b) / 2 in the input's precision
    mono /= 2
    mono *= loudest / _peak(mono)
    return mono


def mix_tracks_peak_normalize(audio, sr, stem_db_red=0.0, out_db=0.0):
    """Every channel is scaled to unit peak - the stem (channel 1) `stem_db_red` dB lower than the context - then the
    sum is brought to a peak of `out_db` dBFS (behaviour of mix_functions.py:230-250, including where it computes in
    f64: the two gains are numpy f64 scalars there, so the stem peak and the final scaling round once from f64)."""
    if audio.ndim != 2:
        raise AssertionError("expected audio of shape [n_samples, channels]")
    out_gain = np.power(10.0, out_db / 20.0)
    stem_gain = np.power(10.0, stem_db_red / 20.0)
    if audio.shape[1] < 2:
        mono = np.array(audio[:, 0], copy=True)
    elif audio.shape[1] != 2:
        peaks = np.abs(audio).max(0, keepdims=True)
        peaks[0, 1] *= stem_gain
        mono = (audio / peaks).sum(1)
    else:
        context, stem = audio[:, 0], audio[:, 1]
        peak_c = _peak(context)
        peak_s = audio.dtype.type(np.float64(_peak(stem)) * stem_gain)    # `peaks[0, 1] *= stem_gain`: f64 product, stored back
        mono = context / peak_c
        mono += stem / peak_s
    gain = out_gain / _peak(mono)              # numpy f64 scalar, as in the reference
    np.multiply(mono, gain, out=mono, casting="same_kind")                # `mix *= gain`: f64 product, rounded once
    return mono


MIX_FUNCTIONS = dict(
    PP=mix_tracks_peak_preserve,
    P0=partial(mix_tracks_peak_normalize, stem_db_red=-0, out_db=-3),
    P1=partial(mix_tracks_peak_normalize, stem_db_red=-3, out_db=-3),
    P2=partial(mix_tracks_peak_normalize, stem_db_red=-6, out_db=-3),
)
# Names the reference also registers (BS.1770 loudness + limiter, mix_functions.py:281-344).  They need pyloudnorm and
# numpy_audio_limiter, which sit outside this build: asking for one is an error at
# construction time, never in the middle of a stream.
LOUDNESS_MIXERS = ("L0", "L1", "L2")
DEFAULT_MIX_FUNCTION = "L0"


def resolve_mix_function(name=None, needed=True):
    """Registry lookup used by ``AudioMetrics``.  `needed` = the configuration mixes at all (APA requested).
    A loudness mixer asked for BY NAME is an error right here.  The reference's DEFAULT (mix_function=None -> "L0") is
    different: `AudioMetrics()` must stay constructible - e.g. to load a state file written elsewhere - so the default
    resolves to a stub that raises the same message at the first window it is asked to mix, i.e. at the very start of an
    add_reference() / evaluate() call, never in the middle of a stream."""
    explicit = name is not None
    if name is None:
        name = DEFAULT_MIX_FUNCTION
    if name in MIX_FUNCTIONS:
        return MIX_FUNCTIONS[name]
    if name not in LOUDNESS_MIXERS:
        raise ValueError(f"Unknown mix_function {name}, must be one of {list(MIX_FUNCTIONS) + list(LOUDNESS_MIXERS)}")
    message = (f"mix_function {name!r} is a BS.1770 loudness mixer (pyloudnorm + numpy_audio_limiter), which this build "
               "does not provide; pass mix_function='P0' (peak based) or your own callable f(audio[n, 2], sr) -> audio[n]")
    if needed and explicit:
        raise ValueError(message)

    def unavailable(audio, sr):
        raise ValueError(message)
    return unavailable
