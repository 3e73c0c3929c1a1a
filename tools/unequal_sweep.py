#!/usr/bin/env python3
"""The shape most evaluations have: a LARGE reference set (statistics and radii cached on it after the first call) against a
SMALL candidate set.  Warm evaluate (FAD + KD + PRDC through AudioMetricsData and the metric functions) for 100 000 reference
rows against 1 000 ... 50 000 candidate rows, per metric, with the kernel forms taken."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import audio_metrics_amd as am  # noqa: E402
from audio_metrics_amd import hip_ops as ops  # noqa: E402

d = int(os.environ.get("AB_DIM", "512"))
n_ref = int(os.environ.get("AB_ROWS", "100000"))
gen = torch.Generator(device="cuda").manual_seed(0)
ref = am.AudioMetricsData(True)
ref.add(torch.randn(n_ref, d, generator=gen, device="cuda"))


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out


for k in (5, 10):
    ref.get_radii(k)
for n_cand in [int(v) for v in os.environ.get("AB_CANDS", "300,1000,3000,5000,10000,20000,50000").split(",")]:
    x = torch.randn(n_cand, d, generator=gen, device="cuda") * 1.05 + 0.05
    for k in (5, 10):
        def fresh():
            c = am.AudioMetricsData(True)
            c.add(x)
            return c
        t_add, cand = timed(fresh)
        t_fad, _ = timed(lambda: am.frechet_distance(cand, ref))
        t_kd, _ = timed(lambda: am.kernel_distance(cand, ref))

        def prdc_cold_candidate():
            cand.radii.clear()
            return am.prdc(ref, cand, k)
        t_prdc, res = timed(prdc_cold_candidate)
        print(f"ref {n_ref} x {d}, cand {n_cand:6d}, k={k:2d}: add {t_add:6.3f}  fad {t_fad:6.3f}  kd {t_kd:6.3f}  prdc {t_prdc:7.3f} ms   "
              f"(knn_path cand {ops.knn_path(n_cand, n_cand, d, k)}, prdc_path {ops.prdc_path(n_ref, n_cand, d)})  precision {res['precision']:.4f}", flush=True)
