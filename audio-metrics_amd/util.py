"""Small host-side stream helpers of the front end (window slicing, buffered shuffle).

They sit upstream of the N x D boundary and run on the host.  What has to be preserved from the reference
(src/audio_metrics/util/audio.py:1-14, util/shuffle.py:5-86) is observable behaviour only: which windows come out of an
item, and - because APA's misaligned pairs are drawn through it - the exact order a shuffled stream is emitted in for a
given state of the random generator (one ``randrange`` per evicted item, one final ``shuffle``).
``tests/test_util_cpu.py`` pins both against sequences produced by the reference's functions."""
import random
from collections import deque


def window_starts(n_samples, win_len, hop_len):
    """Start offsets of the complete windows of `win_len` samples, `hop_len` apart, inside `n_samples`."""
    if hop_len <= 0:
        raise ValueError("the hop between windows must be at least one sample")
    n_windows = (n_samples - win_len) // hop_len + 1 if n_samples >= win_len else 0
    return [w * hop_len for w in range(n_windows)]


def audio_slicer(item, win_dur, sr, hop_dur=None, drop_last=True):
    """Fixed-size windows of `win_dur` seconds, hop = window unless `hop_dur` is given; the trailing partial window is
    dropped (or, with drop_last=False, the window shrinks to the item length)."""
    win_len = int(sr * win_dur) if drop_last else min(int(sr * win_dur), len(item))
    hop_len = int(sr * hop_dur) if hop_dur is not None else win_len
    for start in window_starts(len(item), win_len, hop_len):
        yield item[start:start + win_len]


def multi_audio_slicer(items, win_dur, sr, hop_dur=None, drop_last=True):
    for item in items:
        yield from audio_slicer(item, win_dur, sr, hop_dur, drop_last)


class AgingShuffleBuffer:
    """A pool of `capacity` items that releases one item per item taken in.

    The slots wait in a queue ordered by the time of their last refill, oldest first.  An incoming item evicts the
    occupant of a slot drawn uniformly from the `n_eligible` oldest queue positions; the queue's head slot takes the
    vacated position and the refilled slot goes to the back - so a slot cannot be drawn again before
    ``capacity - n_eligible`` further items have arrived.  The reference keeps the same state as an index permutation with a
    rotating start offset (shuffle.py:34-76); ``physical_order`` reproduces that list, which is what its final
    ``rng.shuffle`` permutes when the input ends."""

    def __init__(self, items, min_age, rng):
        self.slots = list(items)
        self.rng = rng
        self.queue = deque(range(len(self.slots)))
        self.n_eligible = len(self.slots) - min(min_age, len(self.slots) - 1) if self.slots else 0
        self.evictions = 0

    def exchange(self, item):
        """Take `item` in, hand the evicted one back."""
        pos = self.rng.randrange(self.n_eligible)
        slot = self.queue[pos]
        self.queue[pos] = self.queue[0]
        self.queue.popleft()
        self.queue.append(slot)
        self.evictions += 1
        released, self.slots[slot] = self.slots[slot], item
        return released

    def physical_order(self):
        """The queue as the reference's list holds it: queue position t lives at index (evictions + t) mod capacity."""
        size = len(self.slots)
        order = [0] * size
        for t, slot in enumerate(self.queue):
            order[(self.evictions + t) % size] = slot
        return order

    def drain(self):
        order = self.physical_order()
        self.rng.shuffle(order)
        return [self.slots[slot] for slot in order]


def shuffle_stream(iterator, buffer_size=100, seed=None, min_age=0, desc=None):
    """Buffered shuffle with a minimum residence age (util/shuffle.py:5-86): the first `buffer_size` items fill the pool,
    every later item releases one (``AgingShuffleBuffer.exchange``), the rest leave in shuffled order when the input ends.
    With seed=None the GLOBAL `random` module is used, like the reference (``random.seed(s)`` makes APA's misaligned pairs
    reproducible).  `desc` (the reference's progress-bar label) is accepted and ignored."""
    source = iter(iterator)
    head = []
    for item in source:
        head.append(item)
        if len(head) >= buffer_size:
            break
    if not head or buffer_size <= 0:
        return
    pool = AgingShuffleBuffer(head, min_age, random if seed is None else random.Random(seed))
    for item in source:
        yield pool.exchange(item)
    yield from pool.drain()
