#!/usr/bin/env python3
"""A/B helper: time am_knn_radii_f32 (and optionally prdc_counts) at the BASELINE size for the engine
variant / workgroup target given in the environment (AM_ENGINE_VARIANT, AM_WG_TARGET) and print a
checksum of the radii so variants can be compared bit for bit."""
import hashlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from audio_metrics_amd import hip_ops as ops  # noqa: E402

n = int(os.environ.get("AB_ROWS", "100000"))
d = int(os.environ.get("AB_DIM", "512"))
k = int(os.environ.get("AB_K", "5"))
reps = int(os.environ.get("AB_REPS", "3"))
from ab_data import make  # noqa: E402
x = make(os.environ.get("AB_DATA", "randn"), n, d, int(os.environ.get("AB_SEED", "0")))
r = ops.knn_radii(x, k)
torch.cuda.synchronize()
ts = []
for _ in range(reps):
    t0 = time.perf_counter()
    r = ops.knn_radii(x, k)
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
h = hashlib.sha1(r.cpu().numpy().tobytes()).hexdigest()[:12]
best = min(ts)
print(f"variant={os.environ.get('AM_ENGINE_VARIANT', '0')} wg_target={os.environ.get('AM_WG_TARGET', '2048')} "
      f"N={n} D={d} k={k}: best {best * 1e3:.2f} ms  median {sorted(ts)[len(ts) // 2] * 1e3:.2f} ms  "
      f"{2 * n * n * d / best / 1e12:.1f} TF  radii sha1 {h}", flush=True)
