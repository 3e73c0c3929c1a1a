"""Stream helpers of the front end against sequences produced by the reference's own functions (tests/golden/util.npz,
written by make_goldens.py `util` from util/shuffle.py:5-86 and util/audio.py:1-14): APA's misaligned pairs are drawn through
shuffle_stream, so its emitted ORDER for a given generator state - and the state it leaves the global generator in - are
part of the drop-in contract."""
import random

import numpy as np

from audio_metrics_amd.util import AgingShuffleBuffer, audio_slicer, multi_audio_slicer, shuffle_stream


def _cases(golden, prefix):
    g = golden("util")
    return g, sorted({key.split("/")[1] for key in g.files if key.startswith(prefix + "/")})


def test_shuffle_stream_emits_the_reference_order(golden):
    g, names = _cases(golden, "stream")
    assert len(names) >= 7
    for name in names:
        gseed, seed, n, buf, age = (int(v) for v in g[f"stream/{name}/params"])
        if gseed >= 0:
            random.seed(gseed)
        got = list(shuffle_stream(iter(range(n)), buffer_size=buf, seed=None if seed < 0 else seed, min_age=age))
        assert got == g[f"stream/{name}/order"].tolist(), name
        if gseed >= 0:                                   # same number of draws from the global generator
            assert random.random() == float(g[f"stream/{name}/next_draw"][0]), name


def test_shuffle_stream_is_a_permutation_that_respects_the_age():
    rng = random.Random(4)
    pool = AgingShuffleBuffer(range(10), min_age=6, rng=rng)
    last_in = {}
    for step, item in enumerate(range(10, 400)):
        out = pool.exchange(item)
        assert step - last_in.get(out, -100) > 6          # an item stays for at least min_age further arrivals
        last_in[item] = step
    rest = pool.drain()
    assert sorted(rest) == sorted(pool.slots) and len(rest) == 10
    assert sorted(list(shuffle_stream(iter(range(77)), buffer_size=13, seed=1, min_age=5))) == list(range(77))


def test_audio_slicer_windows(golden):
    g, names = _cases(golden, "slicer")
    assert len(names) >= 6
    for name in names:
        n, win, sr, hop, drop = g[f"slicer/{name}/params"]
        item = np.arange(int(n), dtype=np.int64)
        wins = list(audio_slicer(item, float(win), int(sr), hop_dur=None if hop < 0 else float(hop), drop_last=bool(drop)))
        got = np.asarray([[w[0], len(w)] for w in wins], dtype=np.int64).reshape(-1, 2)
        assert np.array_equal(got, g[f"slicer/{name}/windows"]), name
    both = list(multi_audio_slicer([np.arange(10), np.arange(7)], 0.5, 8))
    assert [len(w) for w in both] == [4, 4, 4] and both[2][0] == 0


def _pairs(n, length, dtype):
    rng = np.random.default_rng(3)
    if np.issubdtype(dtype, np.integer):
        return [rng.integers(-2000, 2000, size=(length, 2)).astype(dtype) for _ in range(n)]
    return [rng.standard_normal((length, 2)).astype(dtype) for _ in range(n)]


def _collect(batches):
    tags, rows = [], []
    for batch in batches:
        tags += batch["category"].tolist()
        rows += [row.copy() for row in batch["audio"]]
    return tags, rows


def test_rendered_batches_take_the_dtype_of_the_mixed_window():
    """A mix function may return another dtype than its input (pyloudnorm-style mixers return float64 for float32 windows,
    the peak mixers return float for integer PCM): the worker-filled batch ring must carry what the mixer returned, exactly
    like batches_of + np.stack (and the reference's np.stack, embed.py:218-225) does."""
    from audio_metrics_amd.embed import WindowSource, batches_of, rendered_batches
    from audio_metrics_amd.mix_functions import MIX_FUNCTIONS

    def to_f64(audio, sr=None):
        return audio.astype(np.float64).sum(axis=1) * 0.5

    for mix, dtype, want in ((to_f64, np.float32, np.float64), (MIX_FUNCTIONS["P0"], np.int16, None)):
        songs = _pairs(7, 40, dtype)                               # 7 songs of 40 samples, windows of 8: 35 windows
        make = lambda workers: WindowSource(songs, 16, 0.5, mix, apa_mode="candidate", stems_mode=False,  # noqa: E731
                                            mix_workers=workers)
        plain_tags, plain_rows = _collect(batches_of(make(1), batch_size=4))
        for workers in (1, 3):
            tags, rows = _collect(rendered_batches(make(workers), batch_size=4, ring=6))
            assert tags == plain_tags and len(rows) == len(plain_rows) == 35
            for a, b in zip(rows, plain_rows):
                assert a.dtype == b.dtype and np.array_equal(a, b)
            if want is not None:
                assert rows[0].dtype == want


def test_rendered_batches_name_the_shapes_when_a_window_differs():
    import pytest
    from audio_metrics_amd.embed import WindowSource, rendered_batches
    calls = []

    def ragged(audio, sr=None):
        calls.append(1)
        return audio[:, 0] if len(calls) < 3 else audio[:-1, 0]

    source = WindowSource(_pairs(2, 40, np.float32), 16, 0.5, ragged, apa_mode="candidate", mix_workers=1)
    with pytest.raises(ValueError, match=r"\(8,\).*\(7,\)"):
        list(rendered_batches(source, batch_size=4, ring=6))


def test_device_indices_follow_the_reference_rule(monkeypatch):
    """util/gpu_parallel.py:24-28 + audio_metrics.py:276-279: None = every visible GPU, a non-empty list = exactly those, an
    empty one = no replica handler (the embedder runs where it lives); no GPU at all = RuntimeError.  (Host logic: the
    CUDA queries are replaced, nothing touches a device.)"""
    import pytest
    import torch
    from audio_metrics_amd.audio_metrics import _visible_devices
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 4)
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 2)
    cuda = lambda i: torch.device("cuda", i)                                     # noqa: E731
    assert _visible_devices(None) == [cuda(2), cuda(0), cuda(1), cuda(3)]         # the current device is the home of the statistics
    assert _visible_devices([1, 3]) == [cuda(1), cuda(3)]
    assert _visible_devices(()) == [cuda(2)]

    class OnGpu3:
        def get_device(self):
            return torch.device("cuda", 3)
    assert _visible_devices([], embedder=OnGpu3()) == [cuda(3)]
    # one process per GPU: the launcher gave this rank its device
    assert _visible_devices(None, one_process_per_gpu=True) == [cuda(2)]
    monkeypatch.setattr(torch.cuda, "is_available", lambda: False)
    with pytest.raises(RuntimeError, match="No GPUs found"):
        _visible_devices(None)
