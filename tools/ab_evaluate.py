#!/usr/bin/env python3
"""PRDC part of one evaluate (radii of both sets + membership counts) at the BASELINE size, wall time per repetition and a
checksum of the outputs: the figure a knob of the dev build has to move.  AB_ROWS / AB_DIM / AB_K / AB_DATA / AB_REPS."""
import hashlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import inputs as gi  # noqa: E402
from audio_metrics_amd import hip_ops as ops  # noqa: E402

n, d, k = (int(os.environ.get(key, dflt)) for key, dflt in (("AB_ROWS", "100000"), ("AB_DIM", "512"), ("AB_K", "5")))
reps = int(os.environ.get("AB_REPS", "5"))
ref, cand = (torch.as_tensor(a).cuda() for a in gi.bench_pair(os.environ.get("AB_DATA", "randn"), n, d))
ops.filter_stats_enable("cuda:0", True)


def once():
    r_ref, r_cand = ops.knn_radii(ref, k), ops.knn_radii(cand, k)
    return (r_ref, r_cand) + tuple(ops.prdc_counts(ref, cand, r_ref, r_cand))


out = once()
torch.cuda.synchronize()
ops.filter_stats_read("cuda:0")
t0 = time.perf_counter()
for _ in range(reps):
    out = once()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / reps * 1e3
stats = ops.filter_stats_read("cuda:0")
digest = hashlib.sha1(b"".join(t.cpu().numpy().tobytes() for t in out)).hexdigest()[:12]
print(f"N={n} D={d} k={k}: {ms:.3f} ms per PRDC pass | queued knn {stats['knn_queued'] / (2 * reps):.0f} cross {stats['prdc_queued'] / reps:.0f} "
      f"fallback {stats['knn_fallback_rows']}/{stats['prdc_fallback_calls']} | sha1 {digest}", flush=True)
