#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own hot-path code.

Runs only in the build container (needs /root/reference).  The reference's
modules are imported in place, unmodified, through a bare package stub that
skips ``audio_metrics/__init__.py`` (which would pull in soxr/pyloudnorm/...,
absent here).  Only inputs' seeds and the reference's OUTPUTS are stored; no
reference source is copied.  Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_goldens.py
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import inputs as gi  # noqa: E402

REF = "/root/reference/src/audio_metrics"
pkg = types.ModuleType("audio_metrics")
pkg.__path__ = [REF]
sys.modules["audio_metrics"] = pkg
r_data = importlib.import_module("audio_metrics.data")
r_fad = importlib.import_module("audio_metrics.metrics.fad")
r_kd = importlib.import_module("audio_metrics.metrics.kd")
r_prdc = importlib.import_module("audio_metrics.metrics.prdc")
r_apa = importlib.import_module("audio_metrics.metrics.apa")

VERSIONS = f"torch {torch.__version__}; numpy {np.__version__}; threads {torch.get_num_threads()}"


def amd(emb, store=True, splits=None):
    d = r_data.AudioMetricsData(store_embeddings=store)
    t = torch.as_tensor(emb)
    if splits is None:
        d.add(t)
    else:
        s = 0
        for b in splits:
            d.add(t[s:s + b])
            s += b
        assert s == len(t)
    return d


def cov_summary(cov):
    c = cov.numpy()
    return dict(trace=np.trace(c), fro=np.linalg.norm(c), block=c[:16, :16].copy(),
                sample=c[::37, ::41].copy())


def gen_stats():
    out = {"versions": VERSIONS}
    for name, (seed, d, splits) in gi.STATS_CASES.items():
        x = gi.randn(seed, sum(splits), d, 1.3, 0.2)
        a = amd(x, store=True, splits=splits)
        out[f"{name}/n"] = a.n
        out[f"{name}/mean"] = a.mean.numpy()
        if d <= 128:
            out[f"{name}/cov"] = a.cov.numpy()
        for k, v in cov_summary(a.cov).items():
            out[f"{name}/cov_{k}"] = v
        a.recompute_stats()
        out[f"{name}/re_mean"] = a.mean.numpy()
        out[f"{name}/re_cov_shape"] = np.array(a.cov.shape)
        if tuple(a.cov.shape) == (d, d):
            for k, v in cov_summary(a.cov).items():
                out[f"{name}/re_cov_{k}"] = v
    np.savez_compressed(os.path.join(HERE, "stats.npz"), **out)


def fad_exact_f64(ref, cand):
    """The quantity fad.py:28-31 defines, evaluated without the reference's f32 statistics: f64 mean / covariance of the
    f32 inputs, tr sqrt(Sx Sy) through the symmetric PSD form sqrt(Sx) Sy sqrt(Sx) (eigvalsh, negative rounding noise
    clamped).  |reference - this| is the reference's own numerical noise on the case."""
    def stats(x):
        x = np.asarray(x, dtype=np.float64)
        mu = x.mean(0)
        xc = x - mu
        return mu, xc.T @ xc / max(len(x) - 1, 1)
    (mx, sx), (my, sy) = stats(cand), stats(ref)
    w, v = np.linalg.eigh(sx)
    root = (v * np.sqrt(np.clip(w, 0.0, None))) @ v.T
    lam = np.linalg.eigvalsh(root @ sy @ root)
    tr_sqrt = np.sqrt(np.clip(lam, 0.0, None)).sum()
    return float(((mx - my) ** 2).sum() + np.trace(sx) + np.trace(sy) - 2.0 * tr_sqrt)


def gen_fad():
    out = {"versions": VERSIONS}
    for name, (kind, seed, nr, nc, d) in gi.FAD_CASES.items():
        ref, cand = gi.pair(kind, seed, nr, nc, d)
        a, b = amd(cand, store=False), amd(ref, store=False)
        out[f"{name}/fad"] = r_fad.frechet_distance(a, b)          # (candidate, reference) as audio_metrics.py:257
        out[f"{name}/fad_swapped"] = r_fad.frechet_distance(b, a)
        c = torch.linalg.eigvals(a.cov @ b.cov).sqrt().real.sum().item()
        out[f"{name}/tr_sqrt"] = c
        out[f"{name}/tr_sum"] = (a.cov.trace() + b.cov.trace()).item()
        out[f"{name}/mean_sq"] = (a.mean - b.mean).square().sum().item()
        # the reference's own noise on this case: |its value - the f64 PSD evaluation of the same definition|
        out[f"{name}/fad_exact_f64"] = fad_exact_f64(ref, cand)
        out[f"{name}/ref_noise"] = max(abs(out[f"{name}/fad"] - out[f"{name}/fad_exact_f64"]),
                                       abs(out[f"{name}/fad_swapped"] - out[f"{name}/fad_exact_f64"]))
        print("fad", name, out[f"{name}/fad"], "exact", out[f"{name}/fad_exact_f64"], "noise", out[f"{name}/ref_noise"], flush=True)
    np.savez_compressed(os.path.join(HERE, "fad.npz"), **out)


def gen_fad_spread():
    """How much the REFERENCE's own Frechet distance moves on the rank-deficient cases (fewer rows than dimensions), where its
    f32 torch.cov leaves rounding dust in the null space and eigvals(...).sqrt().real (fad.py:30) sums the square roots of
    the positive part: the same reference code under 1 / 2 / 8 torch threads (other BLAS blocking), both argument orders, and
    with the rows handed over as float64 (what its own test embedder does - the covariance is then computed in f64,
    data.py:44).  min / max are merged into fad.npz; test_fad_vs_golden checks that the device value lies inside."""
    path = os.path.join(HERE, "fad.npz")
    with np.load(path, allow_pickle=False) as g:
        out = {key: g[key] for key in g.files}
    threads0 = torch.get_num_threads()
    for name, (kind, seed, nr, nc, d) in gi.FAD_CASES.items():
        if min(nr, nc) > d:
            continue
        ref, cand = gi.pair(kind, seed, nr, nc, d)
        values = {}
        for threads in (1, 2, 8):
            torch.set_num_threads(threads)
            for dtype in (np.float32, np.float64):
                a, b = amd(cand.astype(dtype), store=False), amd(ref.astype(dtype), store=False)
                values[(threads, np.dtype(dtype).name, "cand_ref")] = r_fad.frechet_distance(a, b)
                values[(threads, np.dtype(dtype).name, "ref_cand")] = r_fad.frechet_distance(b, a)
        torch.set_num_threads(threads0)
        vals = np.array(list(values.values()))
        out[f"{name}/ref_spread_min"], out[f"{name}/ref_spread_max"] = vals.min(), vals.max()
        out[f"{name}/ref_spread_f32"] = np.array([v for k, v in values.items() if k[1] == "float32"])
        out[f"{name}/ref_spread_f64"] = np.array([v for k, v in values.items() if k[1] == "float64"])
        print("fad spread", name, "reference value", float(out[f"{name}/fad"]), "f64 PSD evaluation", float(out[f"{name}/fad_exact_f64"]),
              "reference range", vals.min(), vals.max(), "relative width", (vals.max() - vals.min()) / abs(vals.mean()), flush=True)
        for k, v in sorted(values.items()):
            print("   ", k, v)
    np.savez_compressed(path, **out)


def gen_kd():
    out = {"versions": VERSIONS}
    for name, (kind, seed, n1, n2, d) in gi.KD_CASES.items():
        f2, f1 = gi.pair(kind, seed, n2, n1, d)                      # set 1 = candidate side
        a, b = amd(f1), amd(f2)
        res = r_kd.kernel_distance(a, b)
        out[f"{name}/mean"] = res["kernel_distance_mean"]
        out[f"{name}/std"] = res["kernel_distance_std"]
        # per-subset values: replay the reference's own loop body with its own functions
        m = 1000 if 1000 < min(n1, n2) else max(1, min(n1, n2) // 2)
        rng = np.random.default_rng(1234)
        mmds = np.zeros(100)
        first = None
        for i in range(100):
            i1 = rng.choice(n1, m, replace=False)
            i2 = rng.choice(n2, m, replace=False)
            if first is None:
                first = (i1[:8].copy(), i2[:8].copy())
            mmds[i] = r_kd.kernel_mmd2(f1[i1], f2[i2], r_kd.polynomial_kernel)
        assert float(np.mean(mmds)) == res["kernel_distance_mean"]
        out[f"{name}/mmds"] = mmds
        out[f"{name}/m"] = m
        out[f"{name}/first_idx1"] = first[0]
        out[f"{name}/first_idx2"] = first[1]
        print("kd", name, res)
        if n1 <= 3000:                      # RBF option (kd.py:86-109, 136-140); sigma chosen to give a non-trivial kernel
            for sigma in ((10.0, 3.0) if kind != "unit" else (10.0, 0.5)):
                rr = r_kd.kid_features_to_metric(f1, f2, kernel_type="rbf", kid_sigma=sigma)
                out[f"{name}/rbf_{sigma}/mean"] = rr["kernel_distance_mean"]
                out[f"{name}/rbf_{sigma}/std"] = rr["kernel_distance_std"]
                print("kd rbf", name, sigma, rr)
    np.savez_compressed(os.path.join(HERE, "kd.npz"), **out)


def gen_prdc():
    out = {"versions": VERSIONS}
    for name, (kind, seed, nr, nc, d, k) in gi.PRDC_CASES.items():
        ref, cand = gi.pair(kind, seed, nr, nc, d)
        a, b = amd(ref), amd(cand)
        res = r_prdc.prdc(a, b, k)
        for key, v in res.items():
            out[f"{name}/{key}"] = v
        out[f"{name}/r_ref"] = a.radii[f"radii_{k}"].numpy()
        out[f"{name}/r_cand"] = b.radii[f"radii_{k}"].numpy()
        dist = torch.cdist(a.embeddings, b.embeddings)
        out[f"{name}/col_count"] = (dist < a.radii[f"radii_{k}"][:, None]).sum(dim=0).numpy().astype(np.int32)
        out[f"{name}/row_any"] = (dist < b.radii[f"radii_{k}"][None, :]).any(dim=1).numpy()
        out[f"{name}/row_min"] = dist.min(dim=1)[0].numpy()
        print("prdc", name, res)
    np.savez_compressed(os.path.join(HERE, "prdc.npz"), **out)


def gen_prdc_large():
    """Reference runs at sizes the HIP path serves with its production f16 filter kernels (>= 32768 rows).
    Per case: radii, integer column counts, row flags and the four values of the reference's own prdc()."""
    import gc
    import time
    out = {"versions": VERSIONS}
    for name, (kind, seed, nr, nc, d, k) in gi.PRDC_LARGE_CASES.items():
        t0 = time.time()
        ref, cand = gi.pair(kind, seed, nr, nc, d)
        a, b = amd(ref), amd(cand)
        res = r_prdc.prdc(a, b, k)
        for key, v in res.items():
            out[f"{name}/{key}"] = v
        r_ref, r_cand = a.radii[f"radii_{k}"], b.radii[f"radii_{k}"]
        out[f"{name}/r_ref"] = r_ref.numpy()
        out[f"{name}/r_cand"] = r_cand.numpy()
        gc.collect()
        dist = torch.cdist(a.embeddings, b.embeddings)
        out[f"{name}/col_count"] = (dist < r_ref[:, None]).sum(dim=0).numpy().astype(np.int32)
        out[f"{name}/row_any"] = (dist < r_cand[None, :]).any(dim=1).numpy()
        out[f"{name}/row_cover"] = (dist.min(dim=1)[0] < r_ref).numpy()
        del dist, a, b
        gc.collect()
        print("prdc_large", name, res, f"{time.time() - t0:.0f}s", flush=True)
    np.savez_compressed(os.path.join(HERE, "prdc_large.npz"), **out)


def gen_bench():
    """bench.py's own 2 x 100k x 512 inputs.  FAD and KD come from the reference's functions themselves; PRDC from
    oracle.prdc_blocked (row blocks of the reference's torch calls: its N x N formulation needs 164 GB there) for every
    (data, k) pair the bench line reports.  Incremental: cases already in bench_prdc.npz are kept (6-7 minutes of CPU each)."""
    import time
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    import oracle
    path = os.path.join(HERE, "bench_prdc.npz")
    out = {"versions": VERSIONS}
    if os.path.exists(path):
        with np.load(path, allow_pickle=False) as g:
            out.update({key: g[key] for key in g.files})
    # (kind, width): 512 = the CLAP-shaped headline sets; 128 = VGGish's width (reference embedders/vggish.py:5-33,
    # BASELINE configs[0]'s shape) at the same 100 000 rows - the bench variant `vggish_128`.  Keys of the narrower sets
    # carry the width: "randn_d128/fad", "randn_d128_k5/precision".
    # "randn_d64_f64": float64 rows of width 64 - what n_pca = 64 hands on (projection.py:20-21) - at the headline row count: the
    # bench variant `pca64_f64`; the reference's functions and the blocked oracle then run in float64 throughout
    for kind_name, kind, width in (("randn", "randn", 512), ("clap", "clap", 512), ("randn_d128", "randn", 128),
                                   ("randn_d64_f64", "randn64", 64)):
        ref, cand = gi.pair64("randn", gi.BENCH_SEED, 100000, 100000, width) if kind == "randn64" else gi.bench_pair(kind, 100000, width)
        kind = kind_name
        if f"{kind}/fad" not in out:
            t0 = time.time()
            a, b = amd(cand, store=True), amd(ref, store=True)
            out[f"{kind}/fad"] = r_fad.frechet_distance(a, b)                 # (candidate, reference), audio_metrics.py:257
            kd = r_kd.kernel_distance(a, b)                                    # audio_metrics.py:260
            out[f"{kind}/kernel_distance_mean"] = kd["kernel_distance_mean"]
            out[f"{kind}/kernel_distance_std"] = kd["kernel_distance_std"]
            print("bench", kind, "fad", out[f"{kind}/fad"], kd, f"{time.time() - t0:.0f}s", flush=True)
            del a, b
            np.savez_compressed(path, **out)
        for k in ((5, 10) if width == 512 else (5,)):
            if f"{kind}_k{k}/precision" in out:
                continue
            t0 = time.time()
            res = oracle.prdc_blocked(ref, cand, k, block=4096)
            for key, v in res.items():
                out[f"{kind}_k{k}/{key}"] = v
            print("bench", kind, k, res, f"{time.time() - t0:.0f}s", flush=True)
            np.savez_compressed(path, **out)


from make_goldens_inputs import mix_inputs  # noqa: E402


def gen_mix():
    """The reference's peak-based mix functions (mix_functions.py:209-250, registry :335-344) on seeded inputs.  Its module
    imports pyloudnorm / numpy_audio_limiter / numba at the top (absent here): environment stubs only, the loudness
    mixers are never called."""
    import importlib.util
    for name in ("pyloudnorm", "numpy_audio_limiter", "opt_einsum", "numba"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["numba"].jit = lambda *a, **k: (lambda f: f)
    sys.modules["pyloudnorm"].Meter = object
    spec = importlib.util.spec_from_file_location("ref_mix_functions", os.path.join(REF, "mix_functions.py"))
    r_mix = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(r_mix)
    out = {"versions": VERSIONS}
    for case, audio in mix_inputs():
        for name in ("PP", "P0", "P1", "P2"):
            if case == "silent" and name != "PP":
                continue
            out[f"{case}/{name}"] = r_mix.MIX_FUNCTIONS[name](audio.copy(), sr=16000)
    np.savez_compressed(os.path.join(HERE, "mix.npz"), **out)
    print("mix", len(out) - 1, "outputs")


def gen_apa():
    out = {"versions": VERSIONS}
    table = [(1.0, 2.0, 3.0), (2.0, 1.0, 3.0), (1.0, 5.0, 3.0), (5.0, 1.0, 3.0), (-1.0, 2.0, 3.0),
             (1.0, -2.0, 3.0), (1.0, 2.0, -3.0), (0.0, 0.0, 0.0), (-1.0, -1.0, -1.0), (2.0, 2.0, 0.0),
             (0.3, 0.3, 1e-9), (1e-3, 2e-3, 5e-4)]
    out["table_in"] = np.array(table)
    out["table_out"] = np.array([r_apa._apa(*t) for t in table])
    d = 64
    ref = gi.randn(51, 1500, d)
    anti = gi.randn(52, 1500, d, 1.2, 0.3)
    for i, (sc, sh) in enumerate([(1.0, 0.02), (1.1, 0.15), (1.2, 0.3), (1.5, 0.8)]):
        cand = gi.randn(53 + i, 1200, d, sc, sh)
        a, b, c = amd(cand, False), amd(ref, False), amd(anti, False)
        out[f"three_set_{i}/apa"] = r_apa.apa(a, b, c)
        out[f"three_set_{i}/params"] = np.array([sc, sh])
        print("apa", i, out[f"three_set_{i}/apa"])
    np.savez_compressed(os.path.join(HERE, "apa.npz"), **out)


def gen_e2e():
    """Full API (A12/A13) through the reference's own AudioMetrics, with ENVIRONMENT shims only
    (SURVEY appendix A recipe 2): stub modules for packages absent here, one pretend GPU, no model
    cloning, and an in-order stand-in for the thread-pool pump so the row order is reproducible."""
    import random
    for name in ("soxr", "pyloudnorm", "pyloudnorm.util", "pyloudnorm.normalize", "numpy_audio_limiter",
                 "opt_einsum", "numba", "appdirs"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["numba"].jit = lambda *a, **k: (lambda f: f)
    sys.modules["pyloudnorm"].Meter = object
    sys.modules["pyloudnorm"].util = sys.modules["pyloudnorm.util"]
    sys.modules["pyloudnorm"].normalize = sys.modules["pyloudnorm.normalize"]
    sys.modules["appdirs"].user_cache_dir = lambda *a, **k: "/tmp"
    del sys.modules["audio_metrics"]
    for k in [k for k in sys.modules if k.startswith("audio_metrics.")]:
        del sys.modules[k]
    sys.path.insert(0, "/root/reference/src")
    import audio_metrics
    import audio_metrics.embed as r_embed
    import audio_metrics.util.gpu_parallel as r_gp
    torch.cuda.device_count = lambda: 1
    r_gp.clone_model = lambda model, device: model

    def in_order(iterator, target, discard_input=True, **kw):
        for item in iterator:
            res = target(item)
            if discard_input:
                yield res
            else:
                item.update(res)
                yield item
    r_embed.cpu_parallel = in_order

    def in_order_gpu(iterator, model, discard_input=True, **kw):
        # the reference's pump yields batches in completion order (cf.as_completed), which is not
        # reproducible; batch order decides the stored row order and hence KD (kd.py:185-186)
        for item in iterator:
            res = model.forward(item)
            if discard_input:
                yield res
            else:
                item.update(res)
                yield item
    r_embed.gpu_parallel = in_order_gpu

    c = gi.E2E
    out = {"versions": VERSIONS}
    for tag, metrics, n_pca in (("all", ["fad", "kd", "prdc", "apa"], None), ("stems", ["fad", "kd", "prdc"], None),
                                ("apa", ["apa"], None), ("pca", ["fad", "kd", "prdc", "apa"], 8),
                                ("pca2", ["fad", "apa"], 8)):
        random.seed(c["random_seed"])
        am = audio_metrics.AudioMetrics(metrics=metrics, embedder=gi.NumpyEmbedder(c["dim"], c["sr"]),
                                        mix_function=gi.e2e_mix, win_dur=c["win_dur"], n_pca=n_pca)
        ref = gi.e2e_pairs(c["seed"], c["n_ref"], c["seconds"], c["sr"])
        cand = gi.e2e_pairs(c["seed"] + 1, c["n_cand"], c["seconds"], c["sr"], stem_gain=1.3)
        if tag == "stems":
            ref, cand = [x[:, 1] for x in ref], [x[:, 1] for x in cand]
        am.add_reference(ref)
        if tag == "pca2":                  # second reference batch + evaluate twice: incremental partial_fit path
            out[f"{tag}/first_apa"] = am.evaluate(cand)["apa"]
            am.add_reference(gi.e2e_pairs(c["seed"] + 2, 30, c["seconds"], c["sr"], stem_gain=0.8))
        res = am.evaluate(cand)
        for k, v in res.items():
            out[f"{tag}/{k}"] = v
        out[f"{tag}/keys"] = np.array(list(res.keys()))
        for attr in ("stem_reference", "mix_reference", "mix_anti_reference"):
            d = getattr(am, attr)
            if d is not None and d.n is not None:
                out[f"{tag}/{attr}/n"] = d.n
                out[f"{tag}/{attr}/mean"] = d.mean.numpy()
                out[f"{tag}/{attr}/cov"] = d.cov.numpy()
        if tag in ("all", "pca"):
            # a state file WRITTEN BY THE REFERENCE (audio_metrics.py:78-89): the build's load_state must read it.
            # save_state calls self.__getstate__(), which plain objects only have from Python 3.11 on (there it
            # returns the instance __dict__); this container runs 3.10, so supply exactly that default.
            if not hasattr(object, "__getstate__"):
                audio_metrics.AudioMetrics.__getstate__ = lambda self: self.__dict__
            am.save_state(os.path.join(HERE, f"reference_state_{tag}.pt"))
        print("e2e", tag, res)
    np.savez_compressed(os.path.join(HERE, "e2e.npz"), **out)


def gen_pca():
    """The reference's IncrementalPCA wrapper (projection.py:6-46) on its own: first fit, incremental
    update, transform."""
    r_proj = importlib.import_module("audio_metrics.projection")
    out = {"versions": VERSIONS + f"; sklearn {__import__('sklearn').__version__}"}
    x1 = gi.decaying(61, 500, 24, decades=1.5, shift=0.3)
    x2 = gi.decaying(62, 300, 24, decades=1.5, scale=1.2, shift=0.1)
    xt = gi.decaying(63, 40, 24, decades=1.5)
    pca = r_proj.IncrementalPCA(n_components=6)
    for step, x in (("fit1", x1), ("fit2", x2)):
        pca.partial_fit(x.copy())
        for k in ("components_", "mean_", "var_", "singular_values_", "explained_variance_", "explained_variance_ratio_"):
            out[f"{step}/{k}"] = np.asarray(getattr(pca, k), dtype=np.float64)
        out[f"{step}/noise_variance_"] = float(pca.noise_variance_)
        out[f"{step}/n_samples_seen_"] = int(pca.n_samples_seen_)
        out[f"{step}/transform"] = pca.transform(xt).numpy()
    np.savez_compressed(os.path.join(HERE, "pca.npz"), **out)
    print("pca singular values", out["fit1/singular_values_"], out["fit2/singular_values_"])


def gen_f64():
    """float64 rows through the reference's own functions: AudioMetricsData.add / get_radii, prdc, kernel_distance - and, for
    the pca cases, its IncrementalPCA wrapper in front (projection.py:6-46, audio_metrics.py:163-209: fit on the reference
    set, transform both).  Every stage then runs in f64 (data.py:39-44, prdc.py:12-13,34-48, kd.py:112-116)."""
    import time
    r_proj = importlib.import_module("audio_metrics.projection")
    out = {"versions": VERSIONS + f"; sklearn {__import__('sklearn').__version__}"}
    for name, (kind, seed, nr, nc, d, n_pca, k) in gi.F64_CASES.items():
        t0 = time.time()
        ref, cand = gi.pair64(kind, seed, nr, nc, d)
        assert ref.dtype == np.float64
        if n_pca:
            pca = r_proj.IncrementalPCA(n_components=n_pca)
            pca.partial_fit(ref.copy())
            out[f"{name}/components"] = np.asarray(pca.components_, dtype=np.float64)
            out[f"{name}/pca_mean"] = np.asarray(pca.mean_, dtype=np.float64)
            ref, cand = pca.transform(ref).numpy(), pca.transform(cand).numpy()
            assert ref.dtype == np.float64 and ref.shape == (nr, n_pca)
            out[f"{name}/proj_ref_head"] = ref[:64].copy()               # (spot check of the projection itself)
        a, b = amd(ref), amd(cand)
        assert a.embeddings.dtype == torch.float64
        out[f"{name}/mean_ref"] = a.mean.numpy()
        out[f"{name}/cov_ref"] = a.cov.numpy()
        res = r_prdc.prdc(a, b, k)
        for key, v in res.items():
            out[f"{name}/{key}"] = v
        r_ref, r_cand = a.radii[f"radii_{k}"], b.radii[f"radii_{k}"]
        assert r_ref.dtype == torch.float64
        out[f"{name}/r_ref"] = r_ref.numpy()
        out[f"{name}/r_cand"] = r_cand.numpy()
        dist = torch.cdist(a.embeddings, b.embeddings)
        out[f"{name}/col_count"] = (dist < r_ref[:, None]).sum(dim=0).numpy().astype(np.int32)
        out[f"{name}/row_any"] = (dist < r_cand[None, :]).any(dim=1).numpy()
        out[f"{name}/row_cover"] = (dist.min(dim=1)[0] < r_ref).numpy()
        out[f"{name}/row_min_head"] = dist.min(dim=1)[0][:256].numpy()
        del dist
        # kernel distance with the candidate as features_1 (audio_metrics.py:260), per-subset values replayed with the
        # reference's own functions
        kd = r_kd.kernel_distance(b, a)
        out[f"{name}/kd_mean"], out[f"{name}/kd_std"] = kd["kernel_distance_mean"], kd["kernel_distance_std"]
        m = 1000 if 1000 < min(nr, nc) else max(1, min(nr, nc) // 2)
        rng = np.random.default_rng(1234)
        mmds = np.zeros(100)
        for i in range(100):
            i1 = rng.choice(nc, m, replace=False)
            i2 = rng.choice(nr, m, replace=False)
            mmds[i] = r_kd.kernel_mmd2(cand[i1], ref[i2], r_kd.polynomial_kernel)
        assert float(np.mean(mmds)) == kd["kernel_distance_mean"]
        out[f"{name}/mmds"] = mmds
        if nr <= 3000:
            rr = r_kd.kid_features_to_metric(cand, ref, kernel_type="rbf", kid_sigma=3.0)
            out[f"{name}/rbf_mean"], out[f"{name}/rbf_std"] = rr["kernel_distance_mean"], rr["kernel_distance_std"]
        out[f"{name}/fad"] = r_fad.frechet_distance(b, a)
        print("f64", name, res, kd, f"{time.time() - t0:.0f}s", flush=True)
    np.savez_compressed(os.path.join(HERE, "f64.npz"), **out)


# (seed of the global generator or None, explicit seed or None, items, buffer_size, min_age) per case
STREAM_CASES = {
    "global_small": (11, None, 40, 8, 0),
    "global_aged": (5, None, 257, 100, 99),
    "seeded": (None, 3, 1000, 100, 10),
    "seeded_tiny_buffer": (None, 7, 50, 1, 0),
    "short_input": (2, None, 5, 100, 3),
    "age_beyond_buffer": (None, 9, 300, 10, 1000),
    "empty": (1, None, 0, 10, 0),
}
SLICER_CASES = {            # (samples, win_dur, sr, hop_dur, drop_last)
    "exact": (160000, 5.0, 16000, None, True),
    "remainder": (170001, 5.0, 16000, None, True),
    "short_dropped": (1000, 5.0, 16000, None, True),
    "short_kept": (1000, 5.0, 16000, None, False),
    "overlap": (48000, 1.0, 16000, 0.25, True),
    "hop_longer": (100000, 0.5, 16000, 1.7, True),
}


def gen_util():
    """Output sequences of the reference's stream helpers (util/shuffle.py:5-86, util/audio.py:1-14): the emitted order of
    shuffle_stream for a given generator state, the (start, length) of every window of audio_slicer."""
    import importlib.util
    import random
    def load(name, rel):
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    r_shuffle, r_audio = load("ref_shuffle", "util/shuffle.py"), load("ref_audio", "util/audio.py")
    out = {"versions": VERSIONS}
    for name, (gseed, seed, n, buf, age) in STREAM_CASES.items():
        if gseed is not None:
            random.seed(gseed)
        order = list(r_shuffle.shuffle_stream(iter(range(n)), buffer_size=buf, seed=seed, min_age=age))
        out[f"stream/{name}/order"] = np.asarray(order, dtype=np.int64)
        out[f"stream/{name}/params"] = np.asarray([-1 if gseed is None else gseed, -1 if seed is None else seed, n, buf, age],
                                                  dtype=np.int64)
        out[f"stream/{name}/next_draw"] = np.asarray([random.random()])      # the global generator's state afterwards
    for name, (n, win, sr, hop, drop) in SLICER_CASES.items():
        item = np.arange(n, dtype=np.int64)
        wins = list(r_audio.audio_slicer(item, win, sr, hop_dur=hop, drop_last=drop))
        out[f"slicer/{name}/windows"] = np.asarray([[w[0], len(w)] for w in wins], dtype=np.int64).reshape(-1, 2)
        out[f"slicer/{name}/params"] = np.asarray([n, win, sr, -1.0 if hop is None else hop, float(drop)])
    np.savez_compressed(os.path.join(HERE, "util.npz"), **out)
    print("util", len(out) - 1, "arrays")


if __name__ == "__main__":
    which = sys.argv[1:] or ["stats", "fad", "kd", "prdc", "apa", "pca", "mix", "util", "e2e"]
    for w in which:
        globals()[f"gen_{w}"]()
    print("done", VERSIONS)
