#!/usr/bin/env python3
"""Times the float64 entry points (csrc/pairwise_f64.hip, the f64 ladder of rows that come out of the PCA projection or a
float64 embedder) at the shapes such rows have: n_pca = 8 ... 128 columns, 20 000 ... 100 000 rows.  One line per shape:
k-NN radii, membership counts, kernel distance (100 subsets of min(1000, rows / 2)), their rates against the dense f64 matrix
peak (78.6 TFLOP/s, v_mfma_f64_16x16x4_f64) and the same calls on float32 copies of the rows for scale.
AB_SHAPES="20000x8,20000x64,100000x64" overrides the shapes."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from audio_metrics_amd import hip_ops as ops  # noqa: E402
from audio_metrics_amd.metrics.kd import device_subset_indices  # noqa: E402

F64_PEAK = 78.6e12
dev = torch.device("cuda:0")


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


shapes = os.environ.get("AB_SHAPES", "20000x8,20000x64,50000x32,100000x8,100000x64,100000x128")
k = int(os.environ.get("AB_K", "5"))
for shape in shapes.split(","):
    n, d = (int(v) for v in shape.split("x"))
    g = torch.Generator(device=dev).manual_seed(n + d)
    x = torch.randn(n, d, generator=g, device=dev, dtype=torch.float64)
    y = torch.randn(n, d, generator=g, device=dev, dtype=torch.float64) * 1.05 + 0.05
    reps = 3 if n * n * d > 2e11 else 10
    out = {}
    for name, (a, b) in {"f64": (x, y), "f32": (x.float(), y.float())}.items():
        t_knn = timed(lambda: ops.knn_radii(a, k), reps)
        ra, rb = ops.knn_radii(a, k), ops.knn_radii(b, k)
        t_cnt = timed(lambda: ops.prdc_counts(a, b, ra, rb), reps)
        m = min(1000, n // 2)
        i1, i2 = device_subset_indices(n, n, 100, m, 1234, dev)
        t_kd = timed(lambda: ops.kd_poly(b, a, i1, i2, 1.0 / d, 1.0, 3), reps)
        out[name] = (t_knn, t_cnt, t_kd, m)
    (t_knn, t_cnt, t_kd, m), f32 = out["f64"], out["f32"]
    flop = 2.0 * n * n * d
    print(f"{n} x {d}, k = {k}: f64 radii {t_knn:.2f} ms (all pairs / time = {flop / t_knn * 1e3 / F64_PEAK:.2f} x the f64 matrix peak: above 1 = the f16 filter route), "
          f"membership {t_cnt:.2f} ms ({flop / t_cnt * 1e3 / F64_PEAK:.2f}), kernel distance 100 x {m}: {t_kd:.2f} ms "
          f"({100 * 3 * 2.0 * m * m * d / t_kd * 1e3 / F64_PEAK:.2f})  |  float32 rows: {f32[0]:.2f} / {f32[1]:.2f} / {f32[2]:.2f} ms", flush=True)
