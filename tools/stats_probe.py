#!/usr/bin/env python3
"""Calls the statistics entry points alone (column sums + centred scatter of a 100k x 512 set) so that a profiler sees
only their kernels; prints the wall time per call.  AB_ROWS / AB_DIM / AB_REPS."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_metrics_amd import hip_ops as ops  # noqa: E402

n, d, reps = (int(os.environ.get(k, v)) for k, v in (("AB_ROWS", "100000"), ("AB_DIM", "512"), ("AB_REPS", "10")))
x = torch.randn(n, d, device="cuda") * 0.3 + 0.5
mean = ops.colsum(x) / n
cov = ops.scatter(x, mean)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    mean = ops.colsum(x) / n
    cov = ops.scatter(x, mean)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) * 1e3 / reps
ref = torch.cov(x.double().T) * (n - 1)
print(f"{n} x {d}: {ms:.3f} ms per (colsum + scatter); 2ND^2 / t = {2.0 * n * d * d / ms * 1e-9:.1f} TF algorithmic; "
      f"max rel diff vs torch f64 {((cov - ref).abs().max() / ref.abs().max()).item():.2e}")
