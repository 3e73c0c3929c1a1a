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
# Names the reference also registers (BS.1770 loudness + limiter, mix_functions.py:281-344; "L0" is its DEFAULT,
# mix_functions.py:345, audio_metrics.py:281-288).  The measurement and the limiter are third-party code - pyloudnorm's
# BS.1770 meter and numpy_audio_limiter, neither in the reference tree nor installable here - so, like laion_clap in
# embedders.py, they are resolved LAZILY: on a box that has both packages "L0" / "L1" / "L2" work as in the reference
# (the same calls in the same order: channel loudness, stem set relative to the context, mix normalised to -20 LUFS,
# limiter above full scale); on a box without them asking for one by name is an error at construction, never mid-stream.
LOUDNESS_MIXERS = dict(L0=0.0, L1=-3.0, L2=-6.0)         # name -> stem_db_red (out_db = -20)
DEFAULT_MIX_FUNCTION = "L0"


def _loudness_backends():
    """(pyloudnorm, numpy_audio_limiter) or None when either is missing."""
    import importlib
    try:
        return importlib.import_module("pyloudnorm"), importlib.import_module("numpy_audio_limiter")
    except ImportError:
        return None


def mix_tracks_loudness(audio, sr, stem_db_red=-4.0, out_db=-20.0, backends=None):
    """Mix (context, stem) with a fixed LOUDNESS relationship (reference mix_functions.py:281-332): the stem is set
    `stem_db_red` LU relative to the context (BS.1770 integrated loudness), the sum is normalised to `out_db` LUFS and,
    if it still exceeds full scale, passed through the limiter with the reference's settings.  Silent channels as there:
    both silent -> the context, one silent -> the other channel (then normalised like a mix)."""
    import warnings
    pyln, limiter = backends or _loudness_backends() or (None, None)
    if pyln is None:
        raise ValueError("the BS.1770 loudness mixers need the packages pyloudnorm and numpy_audio_limiter")
    if audio.ndim != 2:
        raise AssertionError("audio must be (samples, channels)")
    if audio.shape[1] == 1:
        return audio[:, 0]
    peaks = np.abs(audio).max(0)
    silent = peaks < 1e-5
    if silent.all():
        warnings.warn("Both channels silent")
        return audio[:, 0]
    meter = pyln.Meter(sr)
    measure = getattr(meter, "integrated_loudness_numba", meter.integrated_loudness)     # (the reference's own subclass, if given)
    if silent.any():
        warnings.warn("One channel silent")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        if silent.any():
            mix = audio[:, ~silent][:, 0]
        else:
            context, stem = audio.T
            l_context, l_stem = measure(context), measure(stem)
            target = l_context + stem_db_red
            if not np.isinf(l_stem) and not np.isinf(target):
                stem = pyln.normalize.loudness(stem, l_stem, target)
            mix = context + stem
        l_mix = measure(mix)
        if not np.isinf(l_mix) and not np.isinf(out_db):
            mix = pyln.normalize.loudness(mix, l_mix, out_db)
    if np.max(np.abs(mix)) > 1.0:
        mix = limiter.limit(signal=mix.astype(np.float32).reshape((1, -1)), attack_coeff=0.99, release_coeff=0.99, delay=527,
                            threshold=0.5)[0]
    return mix


def resolve_mix_function(name=None, needed=True):
    """Registry lookup used by ``AudioMetrics``.  `needed` = the configuration mixes at all (APA requested).
    Peak mixers are this build's own; the loudness mixers ("L0", the reference's default, "L1", "L2") resolve through
    pyloudnorm + numpy_audio_limiter when both are importable.  Without them a loudness mixer asked for BY NAME is an error
    right here, while the DEFAULT (mix_function=None) keeps `AudioMetrics()` constructible - e.g. to load a state file - and
    raises the same message at the first window it is asked to mix, i.e. at the very start of an add_reference() /
    evaluate() call, never in the middle of a stream."""
    explicit = name is not None
    if name is None:
        name = DEFAULT_MIX_FUNCTION
    if name in MIX_FUNCTIONS:
        return MIX_FUNCTIONS[name]
    if name not in LOUDNESS_MIXERS:
        raise ValueError(f"Unknown mix_function {name}, must be one of {list(MIX_FUNCTIONS) + list(LOUDNESS_MIXERS)}")
    backends = _loudness_backends()
    if backends is not None:
        return partial(mix_tracks_loudness, stem_db_red=LOUDNESS_MIXERS[name], out_db=-20.0, backends=backends)
    message = (f"mix_function {name!r} is a BS.1770 loudness mixer and needs the packages pyloudnorm and numpy_audio_limiter, "
               "which are not installed; install them, or pass mix_function='P0' (peak based) or your own callable "
               "f(audio[n, 2], sr) -> audio[n]")
    if needed and explicit:
        raise ValueError(message)

    def unavailable(audio, sr):
        raise ValueError(message)
    return unavailable
