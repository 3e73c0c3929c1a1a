#!/usr/bin/env python3
"""Host-side breakdown of one evaluate step (development aid): wall time of each phase with a device
synchronisation after it, so host overheads show up next to the kernel times."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import audio_metrics_amd as am  # noqa: E402
from audio_metrics_amd import hip_ops as ops  # noqa: E402
from audio_metrics_amd.metrics.kd import subset_indices  # noqa: E402
from audio_metrics_amd.distributed import evaluate_sharded, global_stats  # noqa: E402

n = int(os.environ.get("AB_ROWS", "20000"))
d = int(os.environ.get("AB_DIM", "512"))
gen = torch.Generator(device="cuda").manual_seed(0)
ref = torch.randn(n, d, generator=gen, device="cuda")
cand = torch.randn(n, d, generator=gen, device="cuda") * 1.05 + 0.05


def timed(label, fn, reps=3):
    out = None
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    print(f"{label:28s} {min(ts):8.2f} ms (max {max(ts):.2f})", flush=True)
    return out


timed("evaluate_sharded", lambda: evaluate_sharded(ref, cand, nearest_k=5))
r1 = timed("knn_radii(ref)", lambda: ops.knn_radii(ref, 5))
r2 = timed("knn_radii(cand)", lambda: ops.knn_radii(cand, 5))
timed("prdc_counts", lambda: ops.prdc_counts(ref, cand, r1, r2))
timed("subset_indices", lambda: subset_indices(n, n, 100, 1000, 1234))
s1 = timed("global_stats(ref)", lambda: global_stats(ref, n, ops, 1, None))
s2 = timed("global_stats(cand)", lambda: global_stats(cand, n, ops, 1, None))
timed("frechet", lambda: ops.frechet(s2[0], s2[1], s1[0], s1[1]))
print(ops.frechet(s2[0], s2[1], s1[0], s1[1]))
