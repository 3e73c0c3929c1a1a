#!/usr/bin/env python3
"""can the device give the reference's FLOAT32-rows Frechet distance when a set has fewer rows than
dimensions?  The reference's value there contains the square roots of the rounding dust its f32 torch.cov leaves in the null
space (fad.py:30).  CPU part (no GPU needed): covariances accumulated in f32 by OTHER code than torch.cov, through the
reference's own eigvals formula.  GPU part: the same f32-accumulated covariances through the device's Newton-Schulz solve.
`--reference` (build container only: imports /root/reference): the reference's OWN value under different add() batch sizes.

Result (profiles/r6/fad_f32_probe.txt): an f32-accumulated covariance of our own making reproduces the reference's float32
value to 2e-7 ... 7e-7 THROUGH ITS eigvals FORMULA - the dust is statistically stable - but (1) the device's Newton-Schulz solve
does not see the dust at all: it returns the float64-rows value to 1e-7 from either covariance, and (2) the reference itself
moves by 1.1e-5 ... 1.5e-4 with the batch size of its add() calls (32 / 16 / 11 rows per call, the sizes its pipeline
produces, embed.py:231-236) on the two cases where f32 and f64 rows differ by 2.6e-4 / 3.6e-4.  A "matched" value is therefore
not defined to 1e-4 in this regime; frechet_distance warns instead (metrics/fad.py)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import inputs as gi  # noqa: E402

g = np.load(os.path.join(ROOT, "tests", "golden", "fad.npz"))


def eig_formula(mx, cx, my, cy):
    mx, cx, my, cy = (torch.as_tensor(t, dtype=torch.float64) for t in (mx, cx, my, cy))
    return ((mx - my).square().sum() + cx.trace() + cy.trace() - 2 * torch.linalg.eigvals(cx @ cy).sqrt().real.sum()).item()


def stats_f32(x):
    """mean in f64, rows centred and rounded to f32, products accumulated in f32 (a plain f32 matmul), / (n - 1) in f32"""
    x = torch.as_tensor(x)
    m64 = x.double().mean(0)
    xc = x - m64.float()
    return m64, ((xc.T @ xc) / (x.shape[0] - 1)).double()


def stats_f64(x):
    x = torch.as_tensor(x).double()
    m = x.mean(0)
    xc = x - m
    return m, (xc.T @ xc) / (x.shape[0] - 1)


gpu = torch.cuda.is_available()
if gpu:
    from audio_metrics_amd.metrics import fad as dev_fad
for name, (kind, seed, nr, nc, d) in gi.FAD_CASES.items():
    if min(nr, nc) > d:
        continue
    ref, cand = gi.pair(kind, seed, nr, nc, d)
    want32, want64 = float(g[f"{name}/fad"]), float(g[f"{name}/ref_spread_f64"][0])
    a32, b32, a64, b64 = stats_f32(cand), stats_f32(ref), stats_f64(cand), stats_f64(ref)
    line = [f"{name}: reference f32 rows {want32:.9f}, f64 rows {want64:.9f} (apart {abs(want32 - want64) / abs(want32):.1e})",
            f"own f32 covariance + eigvals formula: rel {abs(eig_formula(*a32, *b32) - want32) / abs(want32):.1e}"]
    if gpu:
        for tag, (a, b) in (("own f32 covariance", (a32, b32)), ("f64 covariance", (a64, b64))):
            got = dev_fad._frechet_distance(a[0], a[1], b[0], b[1])
            line.append(f"device NS on {tag}: {got:.9f} rel to f32-rows {abs(got - want32) / abs(want32):.1e}, to f64-rows "
                        f"{abs(got - want64) / abs(want64):.1e} ({dev_fad.last_info.get('iters')} iterations)")
    print("\n    ".join(line), flush=True)


if "--reference" in sys.argv:
    import importlib
    import types
    pkg = types.ModuleType("audio_metrics")
    pkg.__path__ = ["/root/reference/src/audio_metrics"]
    sys.modules["audio_metrics"] = pkg
    r_data = importlib.import_module("audio_metrics.data")
    r_fad = importlib.import_module("audio_metrics.metrics.fad")

    def feed(x, b):
        d = r_data.AudioMetricsData(store_embeddings=False)
        x = torch.as_tensor(x)
        for s in range(0, len(x), b):
            d.add(x[s:s + b])
        return d

    for name, (kind, seed, nr, nc, dd) in gi.FAD_CASES.items():
        if min(nr, nc) > dd:
            continue
        ref, cand = gi.pair(kind, seed, nr, nc, dd)
        vals = {b: r_fad.frechet_distance(feed(cand, b), feed(ref, b)) for b in (10 ** 9, 32, 16, 11)}
        v64 = r_fad.frechet_distance(feed(cand.astype(np.float64), 10 ** 9), feed(ref.astype(np.float64), 10 ** 9))
        base = vals[10 ** 9]
        print(f"reference itself, {name}: one-shot add of f32 rows {base:.9f}; add() in batches of 32 / 16 / 11 rows moves it by",
              " / ".join(f"{abs(vals[b] - base) / abs(base):.2e}" for b in (32, 16, 11)), f"; f64 rows by {abs(v64 - base) / abs(base):.2e}")
