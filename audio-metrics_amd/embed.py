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
import os
import queue
import threading
import time
from collections import deque
from concurrent.futures import ThreadPoolExecutor
from enum import IntEnum
from itertools import tee

import numpy as np
import torch

from .data import AudioMetricsData, ensure_ndarray
from .util import multi_audio_slicer, shuffle_stream

# Optional stage clock (bench.py --config e2e): a dict that the stages below add their seconds to -
#   mix_wait  the consumer waiting for mixed windows      mix_cpu  summed worker time inside the mix functions
#   batch     filling the [batch, n_samples] host arrays  forward  embedder forward (incl. its host-to-device copy)
#   file      routing rows into the per-category sets (enqueue time; the kernels run asynchronously)
PROFILE = None


def _tick(key, t0):
    if PROFILE is not None:
        PROFILE[key] = PROFILE.get(key, 0.0) + (time.perf_counter() - t0)


def default_mix_workers():
    """Host threads of the mix stage.  The reference uses 64 (embed.py:190-201); numpy releases the GIL inside the array
    loops the mixers consist of, so the stage scales until the cores run out."""
    return max(1, min(32, os.cpu_count() or 1))


def ordered_map(fn, items, workers, lookahead):
    """fn(item) for every item on a thread pool, results IN INPUT ORDER, at most `lookahead` items in flight.  (The
    reference's cpu_parallel yields in completion order, util/cpu_parallel.py:26-62; keeping the order makes the stored
    rows - and with them KD / PRDC - reproducible.)  An exception raised by fn surfaces at its item's position."""
    if workers <= 1:
        for item in items:
            yield fn(item)
        return
    with ThreadPoolExecutor(max_workers=workers, thread_name_prefix="am-mix") as pool:
        pending = deque()
        for item in items:
            pending.append(pool.submit(fn, item))
            if len(pending) >= lookahead:
                t0 = time.perf_counter()
                result = pending.popleft().result()
                _tick("mix_wait", t0)
                yield result
        while pending:
            t0 = time.perf_counter()
            result = pending.popleft().result()
            _tick("mix_wait", t0)
            yield result


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
                 song_buffer_size=100, win_buffer_size=1000, win_min_age=100, seed=None, mix_workers=None):
        self.waveforms, self.sr, self.win_dur = waveforms, sr, win_dur
        self.mix_workers = default_mix_workers() if mix_workers is None else int(mix_workers)
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

    def _jobs(self):
        """(category, window, partner) in emission order.  Everything that draws from `random` - the song shuffle and the
        partner shuffle - happens HERE, on the consuming thread, one window after the other, exactly as in a serial run;
        only the arithmetic on the windows goes to the pool."""
        windows = self._windows()
        if self.apa_mode == "reference":
            # a second view of the same window stream, delayed and shuffled, supplies the stems of the misaligned pairs
            windows, delayed = tee(windows)
            partners = shuffle_stream(delayed, buffer_size=self.win_buffer_size, min_age=self.win_min_age,
                                      seed=self.seed, desc="shuffling windows")
            stream = zip(windows, partners)
        else:
            stream = ((w, None) for w in windows)
        want_pairs = self.apa_mode is not None
        for window, partner in stream:
            window = ensure_ndarray(window)
            if want_pairs:
                if window.ndim != 2:
                    raise ValueError(APA_SHAPE_MESSAGE)
                yield ItemCategory.aligned, window, None
                if partner is not None:
                    partner = ensure_ndarray(partner)
                    assert partner.ndim == 2, APA_SHAPE_MESSAGE
                    yield ItemCategory.misaligned, window, partner
            if self.stems_mode:
                yield ItemCategory.stem, window, None

    def _render(self, job):
        category, window, partner = job
        if category == ItemCategory.stem:
            return category, (window[:, -1] if window.ndim == 2 else window)
        t0 = time.perf_counter()
        if partner is not None:
            window = np.column_stack((window[:, 0], partner[:, 1]))               # this context, another window's stem
        mixed = self.mix_function(window, sr=self.sr)
        _tick("mix_cpu", t0)
        return category, mixed

    def __iter__(self):
        if self.apa_mode is None:                                                  # stems only: nothing to compute
            return (self._render(job) for job in self._jobs())
        return ordered_map(self._render, self._jobs(), self.mix_workers, lookahead=2 * self.mix_workers + 2)


def batches_of(tagged_windows, batch_size=32, ring=0, guards=None):
    """Embedder-protocol batches {"audio": [b, n], "category": [b]} of consecutive windows.
    ring = 0: every batch owns a fresh array.  ring = R > 0: the audio arrays are R buffers used in turn - a 32 x 240000
    f32 batch is 30 MB, and a fresh allocation of that size is returned to the OS on free and page-faulted in again on the
    next one (measured: 117 ms per np.stack against 4 ms for the copies) - valid as long as no more than R - 1 batches
    are held downstream at a time (EmbedderPool sizes it from its queue depth).  A ring batch carries its buffer index
    as batch["_slot"]; a consumer whose device copy of the audio may still be in flight when its forward returns puts an
    event into guards[slot], and the buffer is not refilled before that event has completed."""
    buffers, turn = [], 0
    tags, rows = [], []

    def emit():
        nonlocal turn
        t0 = time.perf_counter()
        first = rows[0]
        uniform = all(r.shape == first.shape and r.dtype == first.dtype for r in rows)
        if ring > 0 and uniform and first.ndim == 1:
            if len(buffers) < ring:
                buffers.append(np.empty((batch_size, first.shape[0]), dtype=first.dtype))
            slot = turn % len(buffers)
            turn += 1
            guard = guards.pop(slot, None) if guards is not None else None
            if guard is not None:
                guard.synchronize()
            buf = buffers[slot]
            if buf.shape[1] != first.shape[0] or buf.dtype != first.dtype:
                buf = buffers[slot] = np.empty((batch_size, first.shape[0]), dtype=first.dtype)
            for i, r in enumerate(rows):
                buf[i] = r
            batch = {"audio": buf[:len(rows)], "category": np.array(tags), "_slot": slot}
        else:
            batch = {"audio": np.stack(rows), "category": np.array(tags)}
        _tick("batch", t0)
        return batch

    for tag, samples in tagged_windows:
        tags.append(int(tag))
        rows.append(samples)
        if len(tags) == batch_size:
            yield emit()
            tags, rows = [], []
    if tags:
        yield emit()


def rendered_batches(source, batch_size=32, ring=4, guards=None):
    """``batches_of(source, ...)`` with the copy into the batch buffer done by the worker that mixed the window: a job
    knows its (batch, row) position when it is created - jobs are numbered in emission order - so the consumer thread only
    hands out rows and collects finished batches.  (Copying 32 x 960 KB per batch on the consumer thread was a third of
    the pipeline's wall time once mixing itself ran on the pool.)  The batch buffers take the dtype and length of the first
    RENDERED window (a mix function may return float64 for float32 input, or float for integer PCM - the mixers divide);
    the first window is therefore rendered on the consumer thread.  Stems-only streams (nothing to mix) and mixers whose
    output is not a 1-D array go through ``batches_of``; a later window whose rendered length differs is a ValueError that
    names both shapes (the reference's np.stack fails on it too, embed.py:218-225)."""
    if source.apa_mode is None:
        yield from batches_of(source, batch_size, ring=ring, guards=guards)
        return
    workers = source.mix_workers
    lookahead = 2 * max(workers, 1) + 2
    jobs = iter(source._jobs())
    try:
        first_job = next(jobs)
    except StopIteration:
        return
    first_category, first_samples = source._render(first_job)
    first_samples = np.asarray(first_samples)
    if first_samples.ndim != 1:
        from itertools import chain
        rest = ordered_map(source._render, jobs, workers, lookahead)
        yield from batches_of(chain([(first_category, first_samples)], rest), batch_size, ring=ring, guards=guards)
        return
    out_len, out_dtype = first_samples.shape[0], first_samples.dtype
    buffers = []                                   # ring of [batch_size, out_len] arrays

    def buffer_for(batch_no):
        slot = batch_no % ring
        if slot >= len(buffers):
            buffers.append(np.empty((batch_size, out_len), dtype=out_dtype))
        return slot, buffers[slot]

    def copy_in(samples, dst):
        samples = np.asarray(samples)
        if samples.shape != dst.shape:
            raise ValueError(f"mixed windows of different shapes in one stream: the first one rendered to {dst.shape} "
                             f"({dst.dtype}), a later one to {samples.shape} ({samples.dtype})")
        t0 = time.perf_counter()
        np.copyto(dst, samples, casting="unsafe")      # (a dtype that differs from the first window's is converted to it)
        _tick("batch_copy_in_workers", t0)

    def render_into(job, dst):
        category, samples = source._render(job)
        copy_in(samples, dst)
        return int(category)

    pool = ThreadPoolExecutor(max_workers=max(workers, 1), thread_name_prefix="am-mix")
    try:
        pending = deque()                          # (future or finished tag, batch number, row)
        tags, batch_no = [], 0

        def finish_batch(rows_in_batch, slot):
            batch = {"audio": buffers[slot][:rows_in_batch], "category": np.array(tags[:rows_in_batch]), "_slot": slot}
            del tags[:rows_in_batch]
            return batch

        def drain(limit):
            # collect finished jobs in order until at most `limit` are pending; yields complete batches
            while len(pending) > limit:
                fut, bno, row = pending.popleft()
                t0 = time.perf_counter()
                tags.append(fut if isinstance(fut, int) else fut.result())
                _tick("mix_wait", t0)
                if row == batch_size - 1:
                    yield finish_batch(batch_size, bno % ring)

        row = 0
        from itertools import chain
        for job in chain([None], jobs):            # None = the window already rendered above
            if row == 0:
                # the buffer this batch goes to may still be read by the device copy of the batch that used it `ring` batches ago
                slot = batch_no % ring
                guard = guards.pop(slot, None) if guards is not None else None
                if guard is not None:
                    guard.synchronize()
                # and every job writing to it must have been collected (they have: ring > batches in flight in `pending`)
            slot, buf = buffer_for(batch_no)
            if job is None:
                copy_in(first_samples, buf[row])
                pending.append((int(first_category), batch_no, row))
            else:
                pending.append((pool.submit(render_into, job, buf[row]), batch_no, row))
            row += 1
            if row == batch_size:
                row, batch_no = 0, batch_no + 1
            yield from drain(lookahead)
        yield from drain(0)
        if row:                                    # the last, partial batch
            yield finish_batch(row, batch_no % ring)
    finally:
        pool.shutdown(wait=True, cancel_futures=True)


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

    DEALINGS = ("round_robin", "free")

    def __init__(self, embedder, devices, dealing="round_robin"):
        """dealing: how batches reach the replicas.  "round_robin" (default): batch b goes to replica b mod R - the stored row
        order is a pure function of the input order and R, so the kernel-distance subsets (drawn by row index) are
        reproducible; one slow replica stalls the dealer once its inbox is full.  "free": every replica pulls from ONE shared
        queue, i.e. a batch goes to whichever GPU is free - the reference's scheme (util/gpu_parallel.py:20-76, submit
        :59-76); no stall, and like the reference's the row order differs from run to run."""
        if not devices:
            raise RuntimeError("No GPUs found, cannot compute audio metrics")
        if dealing not in self.DEALINGS:
            raise ValueError(f"dealing must be one of {self.DEALINGS}, got {dealing!r}")
        self.dealing = dealing
        self.batches_per_replica = []                                # of the last multi-replica run (diagnostics, tests)
        self.devices = [torch.device(d) for d in devices]
        self.replicas = [_replica(embedder, d) for d in self.devices]
        self.guards = {}                                             # batch-ring slot -> event behind its last device copy

    QUEUE_DEPTH = 4                                                   # batches queued per GPU ahead of its worker

    def batches_in_flight(self):
        """How many batches can be alive downstream of the batch generator at once (ring size for batches_of)."""
        if len(self.devices) == 1:
            return 2
        return len(self.devices) * (self.QUEUE_DEPTH + 1) + 2

    def _consumed(self, batch, device):
        """The embedder's host-to-device copy of batch["audio"] is stream-ordered on `device`: an event behind it tells
        the batch ring when the host buffer may be refilled."""
        slot = batch.get("_slot")
        if slot is not None and self.guards is not None:
            event = torch.cuda.Event()
            event.record(torch.cuda.current_stream(device))
            self.guards[slot] = event

    def run(self, batches, wanted):
        """-> {ItemCategory: AudioMetricsData} on devices[0]."""
        aggregators = [CategoryAggregator(wanted, d) for d in self.devices]
        if len(self.devices) == 1:
            with torch.cuda.device(self.devices[0]):
                for batch in batches:
                    t0 = time.perf_counter()
                    embedding = self.replicas[0].forward(batch)["embedding"]
                    self._consumed(batch, self.devices[0])
                    _tick("forward", t0)
                    t0 = time.perf_counter()
                    aggregators[0].file(embedding, batch["category"])
                    _tick("file", t0)
            return aggregators[0].data
        self._run_threads(batches, aggregators)
        return merge_across_devices([a.data for a in aggregators], self.devices[0])

    def _run_threads(self, batches, aggregators):
        if self.dealing == "free":                                   # one shared queue: whoever is free takes the next batch
            shared = queue.Queue(maxsize=self.QUEUE_DEPTH * len(self.devices))
            inboxes = [shared] * len(self.devices)
        else:
            inboxes = [queue.Queue(maxsize=self.QUEUE_DEPTH) for _ in self.devices]
        failures = []
        taken = [0] * len(self.devices)

        def worker(slot):
            try:
                with torch.cuda.device(self.devices[slot]):
                    while True:
                        batch = inboxes[slot].get()
                        if batch is None:
                            return
                        if failures:
                            continue                                # drain so the dealer never blocks
                        embedding = self.replicas[slot].forward(batch)["embedding"]
                        self._consumed(batch, self.devices[slot])
                        aggregators[slot].file(embedding, batch["category"])
                        taken[slot] += 1
            except BaseException as e:                              # surfaced by the dealer below
                failures.append(e)
                while inboxes[slot].get() is not None:           # (shared queue: this worker keeps draining until ITS stop mark)
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
            for box in inboxes:                                     # one stop mark per worker (a shared queue gets all of them)
                box.put(None)
            for t in threads:
                t.join()
            self.batches_per_replica = taken
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
                       device=None, mix_workers=None):
    """{ItemCategory: AudioMetricsData} with device-resident statistics (signature of embed.py:93-109).

    `waveforms`: array/tensor (batch, n_samples[, 2]) or any iterable of (n_samples[, 2]) arrays.
    `gpu_handler`: an ``EmbedderPool`` (several GPUs) - the counterpart of the reference's GPUWorkerHandler; without
    one the embedder runs where it lives and the statistics are kept on `device`."""
    source = WindowSource(waveforms, embedder.sr, win_dur, mix_function, apa_mode=apa_mode, stems_mode=stems_mode,
                          input_sr=input_sr, song_buffer_size=song_buffer_size, win_buffer_size=win_buffer_size,
                          win_min_age=win_min_age, seed=seed, mix_workers=mix_workers)
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
    pool.guards = {}
    # ring: batches alive downstream + the batches the mix workers are still writing into (lookahead / batch_size, rounded up)
    ring = pool.batches_in_flight() + 2 + (2 * source.mix_workers + 2 + batch_size - 1) // batch_size
    return pool.run(rendered_batches(source, batch_size, ring=ring, guards=pool.guards), wanted)
