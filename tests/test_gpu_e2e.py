"""End-to-end parity of the front end (A12/A13): AudioMetrics.add_reference / evaluate
against goldens produced by the reference's own AudioMetrics (same synthetic audio, same
host-side numpy embedder and mix function, same random.seed)."""
import random

import numpy as np
import pytest
import torch

import inputs as gi

pytestmark = pytest.mark.gpu
REL = 1e-4


@pytest.fixture(scope="module")
def am():
    import audio_metrics_amd
    audio_metrics_amd._lib.load()
    return audio_metrics_amd


def make(am, metrics, **kw):
    c = gi.E2E
    # one embedder replica unless a test asks otherwise: the goldens were produced by ONE process, and the kernel-distance
    # subsets are drawn by row index - on a box with several GPUs the default (None = every visible GPU) stores another order
    kw.setdefault("device_indices", [0])
    return am.AudioMetrics(metrics=metrics, embedder=gi.NumpyEmbedder(c["dim"], c["sr"]), mix_function=gi.e2e_mix,
                           win_dur=c["win_dur"], **kw)


def data(stems_only=False):
    c = gi.E2E
    ref = gi.e2e_pairs(c["seed"], c["n_ref"], c["seconds"], c["sr"])
    cand = gi.e2e_pairs(c["seed"] + 1, c["n_cand"], c["seconds"], c["sr"], stem_gain=1.3)
    if stems_only:
        ref, cand = [x[:, 1] for x in ref], [x[:, 1] for x in cand]
    return ref, cand


def tol(key, want):
    if key.startswith("kernel_distance"):
        return max(REL * abs(want), 5e-7)
    if key in ("precision", "recall", "coverage", "density"):
        return max(REL * abs(want), 1.0 / 150)          # one membership flip among 150 candidate windows
    return REL * abs(want)


@pytest.mark.parametrize("tag,metrics,n_pca", [("all", ["fad", "kd", "prdc", "apa"], None),
                                               ("stems", ["fad", "kd", "prdc"], None), ("apa", ["apa"], None),
                                               ("pca", ["fad", "kd", "prdc", "apa"], 8), ("pca2", ["fad", "apa"], 8)])
def test_evaluate_matches_reference(am, golden, tag, metrics, n_pca):
    g = golden("e2e")
    random.seed(gi.E2E["random_seed"])
    m = make(am, metrics, n_pca=n_pca)
    ref, cand = data(stems_only=(tag == "stems"))
    m.add_reference(ref)
    if tag == "pca2":                      # second reference batch: the incremental partial_fit path of the PCA
        c = gi.E2E
        first = m.evaluate(cand)["apa"]
        assert abs(first - float(g["pca2/first_apa"])) <= max(REL * abs(float(g["pca2/first_apa"])), 1e-6)
        m.add_reference(gi.e2e_pairs(c["seed"] + 2, 30, c["seconds"], c["sr"], stem_gain=0.8))
    res = m.evaluate(cand)
    assert list(res) == [str(k) for k in g[f"{tag}/keys"]]
    for key, v in res.items():
        want = float(g[f"{tag}/{key}"])
        extra = 1e-6 if key == "apa" else 0.0          # apa can be exactly 0 (clamped)
        assert abs(v - want) <= tol(key, want) + extra, (key, v, want)
        assert isinstance(v, float)
    for attr in ("stem_reference", "mix_reference", "mix_anti_reference"):
        if f"{tag}/{attr}/n" in g.files:
            d = getattr(m, attr)
            assert d.n == int(g[f"{tag}/{attr}/n"])
            np.testing.assert_allclose(d.mean.cpu().numpy(), g[f"{tag}/{attr}/mean"], rtol=1e-5, atol=1e-7)
            np.testing.assert_allclose(d.cov.cpu().numpy(), g[f"{tag}/{attr}/cov"], rtol=1e-4, atol=1e-8)
    if tag == "pca":
        # behind the projection every row is float64 (projection.py:20-21), and since round 5 it stays float64 through the
        # k-NN radii, the membership counts and the kernel distance: the four PRDC values are the reference's to the digit
        shadows = [v for k, v in vars(m).items() if isinstance(v, am.AudioMetricsData) and v.embeddings is not None
                   and v.embeddings.shape[1] == n_pca]
        assert shadows and all(s.embeddings.dtype == torch.float64 for s in shadows)
        assert all(r.dtype == torch.float64 for s in shadows for r in s.radii.values())
        for key in ("precision", "recall", "density", "coverage"):
            assert res[key] == float(g[f"{tag}/{key}"]), (key, res[key], float(g[f"{tag}/{key}"]))
        assert abs(res["kernel_distance_mean"] - float(g[f"{tag}/kernel_distance_mean"])) <= 1e-9
    # a second evaluate() reuses the cached reference side (radii, d_x_xp) and gives the same values
    res2 = m(cand)
    for key in res:
        assert abs(res2[key] - res[key]) <= 1e-9 * max(1.0, abs(res[key]))


@pytest.mark.parametrize("tag,metrics,n_pca", [("all", ["fad", "kd", "prdc", "apa"], None),
                                               ("pca", ["fad", "kd", "prdc", "apa"], 8)])
def test_load_state_written_by_reference(am, golden, tag, metrics, n_pca):
    """N3: a state file written by the REFERENCE's save_state (tests/golden/reference_state_*.pt, produced by
    make_goldens.py) restores the reference side here - statistics, stored embeddings, cached radii and d_x_xp, PCA
    projections - and evaluate() of the same candidates then gives the reference's results."""
    import os
    g = golden("e2e")
    fp = os.path.join(os.path.dirname(os.path.abspath(gi.__file__)), f"reference_state_{tag}.pt")
    m = make(am, metrics, n_pca=n_pca)
    m.load_state(fp)
    assert m.stem_reference.n == int(g[f"{tag}/stem_reference/n"])
    assert m.stem_reference.embeddings.is_cuda and m.stem_reference.mean.is_cuda
    _, cand = data()
    res = m.evaluate(cand)
    assert list(res) == [str(k) for k in g[f"{tag}/keys"]]
    for key, v in res.items():
        want = float(g[f"{tag}/{key}"])
        extra = 1e-6 if key == "apa" else 0.0
        assert abs(v - want) <= tol(key, want) + extra, (key, v, want)


def test_input_containers(am):
    """ndarray (B, n, 2), generator of (n, 2) arrays and torch tensor are all accepted (embed.py:110-147)."""
    ref, cand = data()
    outs = []
    for conv in (lambda x: np.stack(x), lambda x: (a for a in x), lambda x: torch.as_tensor(np.stack(x))):
        random.seed(3)
        m = make(am, ["fad", "apa"])
        m.add_reference(conv(ref))
        outs.append(m.evaluate(conv(cand)))
    for o in outs[1:]:
        assert o.keys() == outs[0].keys()
        for k in o:
            assert abs(o[k] - outs[0][k]) <= 1e-9 * max(1.0, abs(outs[0][k]))


def test_errors(am):
    ref, cand = data(stems_only=True)
    m = make(am, ["fad", "apa"])
    with pytest.raises(ValueError):
        m.add_reference(ref)                       # APA requested but items are 1-D (embed.py:54-56)
    m = make(am, ["fad"])
    with pytest.raises(ValueError):
        m.evaluate(cand)                           # empty reference (audio_metrics.py:300-313)
    m.add_reference([x[:100] for x in ref])        # shorter than win_dur -> still empty
    with pytest.raises(ValueError):
        m.evaluate(cand)
    with pytest.raises(ValueError):
        am.AudioMetrics(metrics=["fad"], embedder="no_such_embedder")
    with pytest.raises(ValueError):
        am.AudioMetrics(metrics=["fad"], embedder=gi.NumpyEmbedder(), mix_function="no_such_mix")


def test_save_load_state_roundtrip(am, tmp_path):
    """Device analogue of the reference's tests/test_audio_metrics.py:175-197 (rel=abs=1e-6)."""
    ref, cand = data()
    random.seed(11)
    m1 = make(am, ["fad", "kd", "prdc", "apa"], n_pca=10)
    m1.add_reference(ref)
    r1 = m1.evaluate(cand)
    fp = tmp_path / "state.pt"
    m1.save_state(fp)
    state = torch.load(fp, weights_only=True)      # loadable with weights_only=True, reference layout
    assert set(state["stem_reference"]) == {"mean", "n", "cov", "store_embeddings", "embeddings", "radii", "dtype"}
    assert not state["stem_reference"]["mean"].is_cuda
    assert state["stem_projection"]["components_"].shape == (10, gi.E2E["dim"])
    m2 = make(am, ["fad", "kd", "prdc", "apa"], n_pca=10)
    m2.load_state(fp)
    r2 = m2.evaluate(cand)
    assert r1.keys() == r2.keys()
    for k in r1:
        assert r2[k] == pytest.approx(r1[k], rel=1e-6, abs=1e-6)
    m2.reset_reference()
    with pytest.raises(ValueError):
        m2.evaluate(cand)


def test_stale_radius_cache_after_a_second_add_reference_is_an_error_not_a_wild_read(am):
    """The reference never invalidates the radii cache on append (data.py:60-66 vs 68-72): after add_reference(A); evaluate();
    add_reference(B) the stem reference holds radii of len(A) rows beside len(A) + len(B) rows, and its prdc() fails on the
    broadcast.  Here the fused one-call evaluate hands raw pointers to the library: it must NOT reuse such a cache (an
    out-of-bounds device read) - the call falls back to the per-metric path, which raises the shape error."""
    ref, cand = data(stems_only=True)
    m = make(am, ["fad", "prdc"])
    m.add_reference(ref)
    first = m.evaluate(cand)
    assert set(first) == {"fad", "precision", "recall", "density", "coverage"}
    k = min(10, len(m.stem_reference))
    assert m.stem_reference.radii["radii_%d" % k].numel() == m.stem_reference.embeddings.shape[0]
    c = gi.E2E
    m.add_reference([x[:, 1] for x in gi.e2e_pairs(c["seed"] + 2, 30, c["seconds"], c["sr"])])
    assert m.stem_reference.embeddings.shape[0] > m.stem_reference.radii["radii_%d" % k].numel()
    with pytest.raises(ValueError, match="radius / embedding shapes do not match"):
        m.evaluate(cand)
    # the library-level guard of the one-call form
    from audio_metrics_amd import hip_ops as ops
    rows = m.stem_reference.embeddings
    with pytest.raises(ValueError, match="radius / embedding shapes do not match"):
        ops.evaluate(rows, rows, ("fad", "prdc"), 5, given_ref={"radii": torch.zeros(7, dtype=torch.float32, device=rows.device)})
    with pytest.raises(ValueError, match="has shape"):
        ops.evaluate(rows, rows, ("fad", "prdc"), 5, given_ref={"mean": torch.zeros(3, dtype=torch.float64, device=rows.device)})


def test_a_mixer_that_changes_the_dtype_runs_through_the_pipeline(am):
    """A callable mix function returning float64 for float32 windows (what a pyloudnorm-style mixer does) used to abort
    add_reference with an exception without a message; the batch ring now takes the dtype of the mixed window."""
    ref, cand = data()
    results = []
    for mix in (gi.e2e_mix, lambda audio, sr=None: gi.e2e_mix(audio, sr).astype(np.float64)):
        random.seed(9)
        m = am.AudioMetrics(metrics=["fad", "apa"], embedder=gi.NumpyEmbedder(gi.E2E["dim"], gi.E2E["sr"]), mix_function=mix,
                            win_dur=gi.E2E["win_dur"])
        m.add_reference(ref)
        results.append(m.evaluate(cand))
    for key in results[0]:
        assert abs(results[0][key] - results[1][key]) <= 1e-6 * max(1.0, abs(results[0][key])), key


@pytest.mark.parametrize("metrics", [["fad"], ["kd"], ["prdc"], ["fad", "kd"], ["kd", "prdc"], ["fad", "prdc"], ["prdc", "fad", "kd"]])
def test_every_metric_subset_gives_the_golden_values(am, golden, metrics):
    """Whatever subset of the stem metrics is asked for - and whichever of them the one-call form covers - each requested
    key comes back, in the reference's key order (audio_metrics.py:254-274), with the value of the reference's own run on
    the same stems (golden `stems`, produced with all three)."""
    g = golden("e2e")
    m = make(am, metrics)
    ref, cand = data(stems_only=True)
    m.add_reference(ref)
    res = m.evaluate(cand)
    order = [k for k in (str(k) for k in g["stems/keys"])
             if (k == "fad" and "fad" in metrics) or (k.startswith("kernel") and "kd" in metrics)
             or (k in ("precision", "recall", "density", "coverage") and "prdc" in metrics)]
    assert list(res) == order
    for key, v in res.items():
        want = float(g[f"stems/{key}"])
        assert abs(v - want) <= tol(key, want), (key, v, want)
