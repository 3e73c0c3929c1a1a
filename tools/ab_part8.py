#!/usr/bin/env python3
"""Development aid: the three calls of rank 0 in an 8-rank partitioned k-NN, a few times (for rocprofv3)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_metrics_amd import hip_ops as ops
n, d, k, g = 100000, 512, 5, int(os.environ.get("AB_WORLD", "8"))
x = torch.randn(n, d, device="cuda", generator=torch.Generator(device="cuda").manual_seed(0))
rows = n // g
bounds = torch.cat([ops.knn_bounds(x, k, p * rows, rows) for p in range(g)])
lists = torch.stack([ops.knn_sym_part(x, k, p, g, bounds) for p in range(g)])
torch.cuda.synchronize()
for _ in range(5):
    ops.knn_bounds(x, k, 0, rows); ops.knn_sym_part(x, k, 0, g, bounds); ops.knn_lists_finish(lists, x, k)
torch.cuda.synchronize()
