"""Audio -> embeddings -> per-category device statistics (the front end feeding the hot path).

Behavioural counterpart of the reference's ``embedding_pipeline`` (src/audio_metrics/embed.py:93-237),
organised as three stages:

  WindowSource        host side: song shuffle -> resample -> window slicing -> (for the APA reference)
                      misaligned partners from a min-age window shuffle -> tagged, mixed, mono windows.
                      The order of the draws from ``random`` is the reference's, so a seeded run pairs the
                      same windows (embed.py:150-201, util/shuffle.py).
  EmbedderPool        one embedder replica per listed GPU, one worker thread per replica, batches dealt
                      round-robin (reference: util/gpu_parallel.py:20-118 hands each batch to whichever GPU
                      is free; the static deal makes the stored row order reproducible).
  CategoryAggregator  per GPU: rows of every embedder output are routed BY CATEGORY ON THE DEVICE into
                      ``AudioMetricsData`` objects living on that GPU (the reference copies each 32-row batch to
                      the host and re-concatenates the stored matrix, embed.py:226-236).  The per-GPU partial
                      statistics are merged at the end (Chan merge across devices = SURVEY 8(e)'s reduction).
"""
import queue
import threading
from enum import IntEnum
from itertools import tee

import numpy as np
import torch

from .data import AudioMetricsData, ensure_ndarray
from .util import multi_audio_slicer, shuffle_stream

APA_SHAPE_MESSAGE = ("When computing APA items should be tensors/arrays of shape [n_samples, 2] "
                     "(pairing context and stem)")


class ItemCategory(IntEnum):
    aligned = 1
    misaligned = 2
    stem = 3


def resample(item, sr_orig, sr_new):
    """Host-side resampling: soxr when importable (the reference's choice, embed.py:69-83), otherwise scipy's
    polyphase resampler (different filter, same protocol - upstream of the hot path)."""
    audio = ensure_ndarray(item)
    try:
        import soxr
    except ImportError:
        from math import gcd
        from scipy.signal import resample_poly
        g = gcd(int(sr_orig), int(sr_new))
        return resample_poly(audio, int(sr_new) // g, int(sr_orig) // g, axis=0)
    return soxr.resample(audio, sr_orig, sr_new)


class WindowSource:
    """Iterable of ``(ItemCategory, mono window)`` for one ``add_reference`` / ``evaluate`` call."""

    def __init__(self, waveforms, sr, win_dur, mix_function, apa_mode=None, stems_mode=False, input_sr=None,
                 song_buffer_size=100, win_buffer_size=1000, win_min_age=100, seed=None):
        self.waveforms, self.sr, self.win_dur = waveforms, sr, win_dur
        self.mix_function = mix_function
        self.apa_mode, self.stems_mode = apa_mode, stems_mode
        self.input_sr = input_sr
        self.song_buffer_size, self.win_buffer_size, self.win_min_age = song_buffer_size, win_buffer_size, win_min_age
        self.seed = seed

    def _windows(self):
        songs = iter(self.waveforms)
        if self.apa_mode == "reference":
            songs = shuffle_stream(songs, buffer_size=self.song_buffer_size, seed=self.seed, desc="shuffling songs")
        if self.input_sr is not None and self.input_sr != self.sr:
            songs = (resample(song, self.input_sr, self.sr) for song in songs)
        return multi_audio_slicer(songs, self.win_dur, sr=self.sr)

    def __iter__(self):
        windows = self._windows()
        want_pairs = self.apa_mode is not None
        if self.apa_mode == "reference":
            # a second view of the same window stream, delayed and shuffled, supplies the stems of the misaligned pairs
            windows, delayed = tee(windows)
            partners = shuffle_stream(delayed, buffer_size=self.win_buffer_size, min_age=self.win_min_age,
                                      seed=self.seed, desc="shuffling windows")
            stream = zip(windows, partners)
        else:
            stream = ((w, None) for w in windows)
        for window, partner in stream:
            window = ensure_ndarray(window)
            if want_pairs:
                if window.ndim != 2:
                    raise ValueError(APA_SHAPE_MESSAGE)
                yield ItemCategory.aligned, self.mix_function(window, sr=self.sr)
                if partner is not None:
                    partner = ensure_ndarray(partner)
                    assert partner.ndim == 2, APA_SHAPE_MESSAGE
                    crossed = np.column_stack((window[:, 0], partner[:, 1]))      # this context, another window's stem
                    yield ItemCategory.misaligned, self.mix_function(crossed, sr=self.sr)
            if self.stems_mode:
                yield ItemCategory.stem, (window[:, -1] if window.ndim == 2 else window)


def batches_of(tagged_windows, batch_size=32):
    """Embedder-protocol batches {"audio": [b, n], "category": [b]} of consecutive windows."""
    tags, audio = [], []
    for tag, samples in tagged_windows:
        tags.append(int(tag))
        audio.append(samples)
        if len(tags) == batch_size:
            yield {"audio": np.stack(audio), "category": np.array(tags)}
            tags, audio = [], []
    if tags:
        yield {"audio": np.stack(audio), "category": np.array(tags)}


class CategoryAggregator:
    """Owns one device's ``AudioMetricsData`` per wanted category and files embedder outputs into them."""

    def __init__(self, wanted, device):
        """wanted: {ItemCategory: store_embeddings flag}."""
        self.device = device
        self.data = {cat: AudioMetricsData(store, device=device) for cat, store in wanted.items()}

    @staticmethod
    def _rows_of(embedding, positions):
        """embedding[positions] without a host<->device round trip when the positions form an arithmetic progression
        (they do: the source interleaves the categories window by window)."""
        if len(positions) == embedding.shape[0]:
            return embedding
        if len(positions) == 1:
            return embedding[int(positions[0]):int(positions[0]) + 1]
        step = int(positions[1] - positions[0])
        if step > 0 and np.all(np.diff(positions) == step):
            return embedding[int(positions[0]):int(positions[-1]) + 1:step]
        return embedding.index_select(0, torch.as_tensor(positions, device=embedding.device))

    def file(self, embedding, category):
        if not isinstance(embedding, torch.Tensor):
            embedding = torch.as_tensor(np.asarray(embedding))
        if embedding.device != self.device:
            embedding = embedding.to(self.device, non_blocking=True)
        category = np.asarray(category)
        for cat, dst in self.data.items():
            positions = np.flatnonzero(category == int(cat))
            if positions.size:
                dst.add(self._rows_of(embedding, positions))


def _replica(embedder, device):
    """The embedder itself when it already lives on `device` (or is not a GPU model), else a deep copy moved
    there (reference: util/gpu_parallel.py:12-17,47-57 round-trips the model through torch.save)."""
    home = embedder.get_device()
    if home.type != "cuda" or (home.index or 0) == device.index:
        return embedder
    import copy
    clone = copy.deepcopy(embedder)
    mover = getattr(clone, "to", None)
    if mover is None:
        raise RuntimeError(f"embedder {type(embedder).__name__} has no .to(device); cannot place a replica on {device}")
    moved = mover(device)
    return clone if moved is None else moved


class EmbedderPool:
    """Runs the embedder forward and the device-side aggregation on every listed GPU."""

    def __init__(self, embedder, devices):
        if not devices:
            raise RuntimeError("No GPUs found, cannot compute audio metrics")
        self.devices = [torch.device(d) for d in devices]
        self.replicas = [_replica(embedder, d) for d in self.devices]

    def run(self, batches, wanted):
        """-> {ItemCategory: AudioMetricsData} on devices[0]."""
        aggregators = [CategoryAggregator(wanted, d) for d in self.devices]
        if len(self.devices) == 1:
            with torch.cuda.device(self.devices[0]):
                for batch in batches:
                    aggregators[0].file(self.replicas[0].forward(batch)["embedding"], batch["category"])
            return aggregators[0].data
        self._run_threads(batches, aggregators)
        return merge_across_devices([a.data for a in aggregators], self.devices[0])

    def _run_threads(self, batches, aggregators):
        depth = 4                                                 # batches queued per GPU ahead of its worker
        inboxes = [queue.Queue(maxsize=depth) for _ in self.devices]
        failures = []

        def worker(slot):
            try:
                with torch.cuda.device(self.devices[slot]):
                    while True:
                        batch = inboxes[slot].get()
                        if batch is None:
                            return
                        if failures:
                            continue                                # drain so the dealer never blocks
                        aggregators[slot].file(self.replicas[slot].forward(batch)["embedding"], batch["category"])
            except BaseException as e:                              # surfaced by the dealer below
                failures.append(e)
                while inboxes[slot].get() is not None:
                    pass

        threads = [threading.Thread(target=worker, args=(s,), name=f"am-embed-{s}", daemon=True)
                   for s in range(len(self.devices))]
        for t in threads:
            t.start()
        try:
            for number, batch in enumerate(batches):
                if failures:
                    break
                inboxes[number % len(inboxes)].put(batch)
        finally:
            for box in inboxes:
                box.put(None)
            for t in threads:
                t.join()
        if failures:
            raise failures[0]


def merge_across_devices(per_device, target):
    """Fold per-GPU {category: AudioMetricsData} partials into one set on `target` (device order = row order)."""
    merged = {}
    for partial in per_device:
        for cat, data in partial.items():
            if cat not in merged:
                merged[cat] = AudioMetricsData(data.store_embeddings, device=target)
            merged[cat] += data.to(target)
    return merged


def embedding_pipeline(waveforms, embedder, mix_function, gpu_handler=None, apa_mode=None, stems_mode=False,
                       store_mix_embeddings=False, store_stem_embeddings=False, batch_size=32, win_dur=5.0,
                       song_buffer_size=100, win_buffer_size=1000, win_min_age=100, seed=None, input_sr=None,
                       device=None):
    """{ItemCategory: AudioMetricsData} with device-resident statistics (signature of embed.py:93-109).

    `waveforms`: array/tensor (batch, n_samples[, 2]) or any iterable of (n_samples[, 2]) arrays.
    `gpu_handler`: an ``EmbedderPool`` (several GPUs) - the counterpart of the reference's GPUWorkerHandler; without
    one the embedder runs where it lives and the statistics are kept on `device`."""
    source = WindowSource(waveforms, embedder.sr, win_dur, mix_function, apa_mode=apa_mode, stems_mode=stems_mode,
                          input_sr=input_sr, song_buffer_size=song_buffer_size, win_buffer_size=win_buffer_size,
                          win_min_age=win_min_age, seed=seed)
    wanted = {}
    if apa_mode is not None:
        wanted[ItemCategory.aligned] = store_mix_embeddings
    if apa_mode == "reference":
        wanted[ItemCategory.misaligned] = store_mix_embeddings
    if stems_mode:
        wanted[ItemCategory.stem] = store_stem_embeddings
    pool = gpu_handler
    if pool is None:
        if device is None:
            from .data import default_device
            home = embedder.get_device()
            device = home if home.type == "cuda" else default_device()
        pool = EmbedderPool(embedder, [device])
    return pool.run(batches_of(source, batch_size), wanted)
