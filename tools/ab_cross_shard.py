#!/usr/bin/env python3
"""Time am_prdc_counts_f32 on a row shard (the per-rank call of a multi-GPU run) against the full candidate set."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_metrics_amd import hip_ops as ops
n, d = 100000, 512
gen = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(n, d, generator=gen, device="cuda"); y = torch.randn(n, d, generator=gen, device="cuda") * 1.05 + 0.05
rx, ry = ops.knn_radii(x, 5), ops.knn_radii(y, 5)
for world in (1, 2, 4, 8):
    rows = n // world
    ops.prdc_counts(x[:rows], y, rx[:rows], ry); torch.cuda.synchronize()
    ts = []
    for _ in range(4):
        t0 = time.perf_counter(); ops.prdc_counts(x[:rows], y, rx[:rows], ry); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"world={world}: shard of {rows} rows: {min(ts) * 1e3:.2f} ms  ({min(ts) * 1e3 * world:.2f} ms x world)")
