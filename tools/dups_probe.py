#!/usr/bin/env python3
"""Sets with a block of IDENTICAL rows (silent windows of a stem dataset embed to the same vector): time of the k-NN radii and of
the membership counts as the share of duplicates grows, with the filter statistics.  AB_ROWS / AB_DIM / AB_K."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from audio_metrics_amd import hip_ops as ops  # noqa: E402

n, d, k = int(os.environ.get("AB_ROWS", "100000")), int(os.environ.get("AB_DIM", "512")), int(os.environ.get("AB_K", "5"))
gen = torch.Generator(device="cuda").manual_seed(0)
for share in [float(v) for v in os.environ.get("AB_SHARES", "0,0.001,0.003,0.01,0.03,0.1").split(",")]:
    x = torch.randn(n, d, generator=gen, device="cuda")
    y = torch.randn(n, d, generator=gen, device="cuda") * 1.05 + 0.05
    m = int(n * share)
    unit = os.environ.get("AB_UNIT", "0") == "1"                   # CLAP-shaped: unit-norm rows; the silent vector too
    if unit:
        x, y = x + 0.5, y + 0.5
    if m:
        silent = torch.randn(1, d, generator=gen, device="cuda") * (1.0 if unit else float(os.environ.get("AB_SILENT_SCALE", "0.1"))) + (0.5 if unit else 0.0)
        x[torch.randperm(n, generator=gen, device="cuda")[:m]] = silent
        y[torch.randperm(n, generator=gen, device="cuda")[:m]] = silent
    if unit:
        x, y = x / x.norm(dim=1, keepdim=True), y / y.norm(dim=1, keepdim=True)
    ops.filter_stats_enable("cuda:0", True)
    out = {}
    for name, fn in (("knn", lambda: ops.knn_radii(x, k)), ("knn_cand", lambda: ops.knn_radii(y, k))):
        out[name] = fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            out[name] = fn()
        torch.cuda.synchronize()
        out[name + "_ms"] = (time.perf_counter() - t0) / 3 * 1e3
    st = ops.filter_stats_read("cuda:0")
    fn = lambda: ops.prdc_counts(x, y, out["knn"], out["knn_cand"])   # noqa: E731
    fn()
    torch.cuda.synchronize()
    ops.filter_stats_read("cuda:0")
    t0 = time.perf_counter()
    for _ in range(3):
        col, rany, rcov = fn()
    torch.cuda.synchronize()
    t_cross = (time.perf_counter() - t0) / 3 * 1e3
    st2 = ops.filter_stats_read("cuda:0")
    print(f"N={n} D={d} k={k} {'unit-norm' if unit else 'randn'} duplicates {100 * share:5.2f} % ({m:6d} rows): knn {out['knn_ms']:8.3f} ms (fallback rows {st['knn_fallback_rows'] / 8:.0f}, "
          f"zero radii {int((out['knn'] == 0).sum())})  membership {t_cross:8.3f} ms (queued {st2['prdc_queued'] / 3:.0f}, fallback calls {st2['prdc_fallback_calls'] / 3:.0f})",
          flush=True)
