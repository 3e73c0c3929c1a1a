#!/usr/bin/env python3
"""Cold evaluate() (FAD + KD + PRDC) over a grid of set sizes, widths and k: one line per shape with the step time and the
kernel forms taken (am_knn_path / am_prdc_path) - to spot path thresholds that sit in the wrong place."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from audio_metrics_amd import hip_ops as ops  # noqa: E402
from audio_metrics_amd.distributed import evaluate_sharded  # noqa: E402

dev = torch.device("cuda:0")
gen = torch.Generator(device="cuda").manual_seed(0)
rows = [int(v) for v in os.environ.get("AB_ROWS_LIST", "1000,4000,8000,8192,16000,32000,32768,50000,100000").split(",")]
dims = [int(v) for v in os.environ.get("AB_DIMS", "128,512").split(",")]
ks = [int(v) for v in os.environ.get("AB_KS", "5,10").split(",")]
for d in dims:
    for n in rows:
        ref = torch.randn(n, d, generator=gen, device=dev)
        cand = torch.randn(n, d, generator=gen, device=dev) * 1.05 + 0.05
        for k in ks:
            evaluate_sharded(ref, cand, nearest_k=k)
            torch.cuda.synchronize()
            reps = 2 if n >= 150000 else 3 if n >= 32000 else 10
            t0 = time.perf_counter()
            for _ in range(reps):
                evaluate_sharded(ref, cand, nearest_k=k)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / reps * 1e3
            ops.filter_stats_enable(dev, True)
            t0 = time.perf_counter()
            for _ in range(reps):
                evaluate_sharded(ref, cand, metrics=("prdc",), nearest_k=k)
            torch.cuda.synchronize()
            ms_prdc = (time.perf_counter() - t0) / reps * 1e3
            st = ops.filter_stats_read(dev)
            ops.filter_stats_enable(dev, False)
            print(f"N={n:7d} D={d:4d} k={k:2d}: evaluate {ms:8.3f} ms  prdc only {ms_prdc:8.3f} ms  ({3 * 2.0 * n * n * d / ms_prdc * 1e-9:7.1f} TF algorithmic)  "
                  f"knn_path {ops.knn_path(n, n, d, k)} prdc_path {ops.prdc_path(n, n, d)}  queued/row {st['knn_queued'] / max(2 * reps * n, 1):6.1f} "
                  f"fallback rows {st['knn_fallback_rows'] / reps:.0f} membership fallbacks {st['prdc_fallback_calls'] / reps:.0f}", flush=True)
