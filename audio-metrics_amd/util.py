"""Small host-side stream helpers of the front end (window slicing, buffered shuffle).

Behavioural restatements of the reference's src/audio_metrics/util/audio.py:1-14 and
util/shuffle.py:5-86; they sit upstream of the N x D boundary and run on the host."""
import random


def audio_slicer(item, win_dur, sr, hop_dur=None, drop_last=True):
    """Fixed-size windows of `win_dur` seconds, hop = window unless `hop_dur` is given;
    the trailing partial window is dropped (or, with drop_last=False, the window shrinks
    to the item length)."""
    n = len(item)
    win_len = int(sr * win_dur)
    if not drop_last:
        win_len = min(win_len, n)
    hop_len = win_len if hop_dur is None else int(sr * hop_dur)
    start = 0
    while start + win_len <= n:
        yield item[start:start + win_len]
        start += hop_len


def multi_audio_slicer(items, win_dur, sr, hop_dur=None, drop_last=True):
    for item in items:
        yield from audio_slicer(item, win_dur, sr, hop_dur, drop_last)


def shuffle_stream(iterator, buffer_size=100, seed=None, min_age=0, desc=None):
    """Buffered shuffle with a minimum residence age (util/shuffle.py:5-86).

    The buffer is filled first; afterwards every incoming item evicts a slot drawn
    uniformly from the `n_eligible = len(buffer) - min(min_age, len(buffer)-1)` slots
    that have waited longest, tracked as a rotating window over an index permutation.
    When the input ends the remaining items are emitted in shuffled order.  With
    seed=None the GLOBAL `random` module is used, exactly like the reference (so
    `random.seed(s)` makes APA's misaligned pairs reproducible); the sequence of
    generator calls (one randrange per evicted item, one final shuffle) is the same."""
    iterator = iter(iterator)
    rng = random if seed is None else random.Random(seed)
    buffer = []
    for _ in range(buffer_size):
        try:
            buffer.append(next(iterator))
        except StopIteration:
            break
    total = len(buffer)
    if total == 0:
        return
    order = list(range(total))
    offset = 0
    n_eligible = total - min(min_age, total - 1)
    for item in iterator:
        j = (offset + rng.randrange(n_eligible)) % total
        slot = order[j]
        yield buffer[slot]
        buffer[slot] = item
        order[j], order[offset] = order[offset], order[j]
        offset = (offset + 1) % total
    rng.shuffle(order)
    for slot in order:
        yield buffer[slot]
