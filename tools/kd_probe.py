#!/usr/bin/env python3
"""Calls the kernel-distance entry point alone (100 subsets of 1000 rows from two 100k x 512 sets) so that a profiler sees
only its kernels; prints the wall time per call."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from audio_metrics_amd import hip_ops as ops  # noqa: E402
from audio_metrics_amd.metrics.kd import subset_indices  # noqa: E402

n, d = int(os.environ.get("AB_ROWS", "100000")), int(os.environ.get("AB_DIM", "512"))
dev = torch.device("cuda:0")
x = torch.randn(n, d, device=dev)
y = torch.randn(n, d, device=dev) * 1.05 + 0.05
idx1, idx2 = subset_indices(n, n, 100, 1000, 1234)
i1, i2 = ops.upload_host_array(idx1, dev), ops.upload_host_array(idx2, dev)
ops.kd_poly(y, x, i1, i2, 1.0 / d, 1, 3)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    r = ops.kd_poly(y, x, i1, i2, 1.0 / d, 1, 3)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) * 100
print(f"kd_poly 100 x 1000 x {d}: {ms:.3f} ms per call; S*3*2*m^2*D / t = {100 * 3 * 2 * 1e6 * d / ms * 1e-9:.1f} TF algorithmic; mean {r.mean().item():.6e}")
