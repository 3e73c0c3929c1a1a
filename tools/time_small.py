#!/usr/bin/env python3
"""Development aid: rocprofv3 target that runs the k-NN entry point a few times (small kernels of the filter path)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_metrics_amd import hip_ops as ops
x = torch.randn(100000, 512, device="cuda", generator=torch.Generator(device="cuda").manual_seed(0))
for _ in range(4): ops.knn_radii(x, 5)
torch.cuda.synchronize()
