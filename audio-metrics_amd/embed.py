"""Audio -> embeddings -> per-category statistics (front end of the hot path).

Mirrors the reference's ``embedding_pipeline`` (src/audio_metrics/embed.py:93-237): same
arguments, same lazy generator chain (song shuffle -> resample -> window slicer ->
tee + min-age window shuffle -> serialize aligned / misaligned / stem items -> mix ->
batches of 32 -> embedder forward -> per-category ``add``), with one deliberate
difference on the hot path: the embedder's output stays on the device.  The reference
copies every 32-row batch to the host (``.cpu()``, embed.py:227) and re-concatenates the
stored embeddings; here category masks are applied on the device and rows go into the
pre-sized HBM buffer of ``AudioMetricsData``.  The embedder runs on ONE GPU as ordinary
PyTorch code; mixing / resampling are plain host code."""
from enum import IntEnum
from functools import partial
from itertools import tee

import numpy as np
import torch

from .data import AudioMetricsData, ensure_ndarray
from .util import multi_audio_slicer, shuffle_stream


class ItemCategory(IntEnum):
    aligned = 1
    misaligned = 2
    stem = 3


def batch_accumulator(items, batch_size=32):
    """Stack consecutive items into {"audio": [b, n], "category": [b]} batches (embed.py:24-41)."""
    audio, category = [], []
    for item in items:
        audio.append(item["audio"])
        category.append(item["category"])
        if len(audio) == batch_size:
            yield {"audio": np.stack(audio), "category": np.array(category)}
            audio, category = [], []
    if audio:
        yield {"audio": np.stack(audio), "category": np.array(category)}


def serialize_items(items1, items2=None, apa_mode=False, stems_mode=False):
    """Per window: the aligned (context, stem) pair, optionally the misaligned pair built
    from this window's context and the shuffled stream's stem, and the stem alone -
    in that order (embed.py:44-66)."""
    pairs = ((it, None) for it in items1) if items2 is None else zip(items1, items2)
    msg = ("When computing APA items should be tensors/arrays of shape [n_samples, 2] "
           "(pairing context and stem)")
    for item1, item2 in pairs:
        item1 = ensure_ndarray(item1)
        if apa_mode:
            if item1.ndim != 2:
                raise ValueError(msg)
            yield {"audio": item1, "category": ItemCategory.aligned}
            if item2 is not None:
                item2 = ensure_ndarray(item2)
                assert item2.ndim == 2, msg
                yield {"audio": np.column_stack((item1[:, 0], item2[:, 1])), "category": ItemCategory.misaligned}
        if stems_mode:
            yield {"audio": item1[:, -1] if item1.ndim == 2 else item1, "category": ItemCategory.stem}


def resample(item, sr_orig, sr_new):
    """Host-side resampling.  The reference uses soxr (embed.py:69-83); it is preferred
    when importable, otherwise scipy's polyphase resampler is used (different filter,
    same protocol - upstream of the hot path)."""
    audio = ensure_ndarray(item)
    try:
        import soxr
        return soxr.resample(audio, sr_orig, sr_new)
    except ImportError:
        from math import gcd
        from scipy.signal import resample_poly
        g = gcd(int(sr_orig), int(sr_new))
        return resample_poly(audio, int(sr_new) // g, int(sr_orig) // g, axis=0)


def mix_pair(data, mix_func, sr):
    if data["category"] == ItemCategory.stem:
        return {"audio": data["audio"]}
    return {"audio": mix_func(data["audio"], sr=sr)}


def embedding_pipeline(waveforms, embedder, mix_function, gpu_handler=None, apa_mode=None, stems_mode=False,
                       store_mix_embeddings=False, store_stem_embeddings=False, batch_size=32, win_dur=5.0,
                       song_buffer_size=100, win_buffer_size=1000, win_min_age=100, seed=None, input_sr=None,
                       device=None):
    """Returns {ItemCategory: AudioMetricsData} with device-resident statistics.

    `waveforms`: array/tensor (batch, n_samples[, 2]) or any iterable of (n_samples[, 2])
    arrays (embed.py:110-147).  `gpu_handler` is accepted for signature compatibility and
    ignored: the embedder forward runs on the embedder's own device."""
    items = iter(waveforms)
    if apa_mode == "reference":
        items = shuffle_stream(items, buffer_size=song_buffer_size, seed=seed, desc="shuffling songs")
    if input_sr is not None and input_sr != embedder.sr:
        items = (resample(it, input_sr, embedder.sr) for it in items)
    items = multi_audio_slicer(items, win_dur, sr=embedder.sr)
    if apa_mode == "reference":
        items, shuffled_items = tee(items)
        shuffled_items = shuffle_stream(shuffled_items, buffer_size=win_buffer_size, min_age=win_min_age, seed=seed,
                                        desc="shuffling windows")
    else:
        shuffled_items = None
    items = serialize_items(items, shuffled_items, apa_mode, stems_mode)
    if apa_mode is not None:
        _mix = partial(mix_pair, mix_func=mix_function, sr=embedder.sr)
        items = ({**item, **_mix(item)} for item in items)
    items = batch_accumulator(items, batch_size=batch_size)

    metrics_data = {}
    if apa_mode is not None:
        metrics_data[ItemCategory.aligned] = AudioMetricsData(store_mix_embeddings, device=device)
    if apa_mode == "reference":
        metrics_data[ItemCategory.misaligned] = AudioMetricsData(store_mix_embeddings, device=device)
    if stems_mode:
        metrics_data[ItemCategory.stem] = AudioMetricsData(store_stem_embeddings, device=device)

    for batch in items:
        embedding = embedder.forward(batch)["embedding"]          # stays on the GPU (no .cpu())
        if not isinstance(embedding, torch.Tensor):
            embedding = torch.as_tensor(np.asarray(embedding))
        category = batch["category"]
        for cat, dst in metrics_data.items():
            rows = np.flatnonzero(category == cat)
            if rows.size:
                if rows.size == len(category):
                    dst.add(embedding)
                else:
                    dst.add(embedding[torch.as_tensor(rows, device=embedding.device)])
    return metrics_data
