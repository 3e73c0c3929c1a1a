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
