#!/usr/bin/env python3
"""Times the two wide f16 filter kernels (knn_wide_kernel, cross_wide_kernel) with the library's kernel clock at the
BASELINE size and prints a checksum of the outputs.  AB_ROWS / AB_DIM / AB_K / AB_DATA (randn | clap) / AB_REPS;
AM_HIP_LIBRARY=dev (or a path to a variant build) selects the library."""
import hashlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import inputs as gi  # noqa: E402
from audio_metrics_amd import hip_ops as ops  # noqa: E402

n, d, k = (int(os.environ.get(key, dflt)) for key, dflt in (("AB_ROWS", "100000"), ("AB_DIM", "512"), ("AB_K", "5")))
reps = int(os.environ.get("AB_REPS", "5"))
ref, cand = (torch.as_tensor(a).cuda() for a in gi.bench_pair(os.environ.get("AB_DATA", "randn"), n, d))
ops.kernel_clock_enable(True)
ops.filter_stats_enable("cuda:0", True)
for rep in range(reps + 1):
    r_ref, r_cand = ops.knn_radii(ref, k), ops.knn_radii(cand, k)
    col, rany, rcov = ops.prdc_counts(ref, cand, r_ref, r_cand)
    if rep == 0:                                              # warm-up launches dropped
        torch.cuda.synchronize()
        for kid in range(4):
            ops.kernel_clock_read(kid)
        ops.filter_stats_read("cuda:0")
torch.cuda.synchronize()
names = ("knn_wide", "cross_wide", "knn_verify", "cross_verify")
clock = {name: ops.kernel_clock_read(kid) for kid, name in enumerate(names)}
stats = ops.filter_stats_read("cuda:0")
digest = hashlib.sha1(b"".join(t.cpu().numpy().tobytes() for t in (r_ref, r_cand, col, rany, rcov))).hexdigest()[:12]
line = " ".join(f"{name} {ms / max(c, 1):.3f} ms" for name, (c, ms) in clock.items())
print(f"[{os.environ.get('AB_TAG', '-')}] N={n} D={d} k={k} {os.environ.get('AB_DATA', 'randn')}: {line} | queued knn {stats['knn_queued'] / (2 * reps):.0f} "
      f"verified {stats['knn_verified_pairs'] / (2 * reps):.0f} cross {stats['prdc_queued'] / reps:.0f} fallback {stats['knn_fallback_rows']}/"
      f"{stats['prdc_fallback_calls']} | sha1 {digest}", flush=True)
