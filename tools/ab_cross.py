#!/usr/bin/env python3
"""A/B helper: time am_prdc_counts_f32 at the BASELINE size (development aid)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from audio_metrics_amd import hip_ops as ops  # noqa: E402

n = int(os.environ.get("AB_ROWS", "100000"))
d = int(os.environ.get("AB_DIM", "512"))
n2 = int(os.environ.get("AB_ROWS2", str(n)))          # candidate rows (default: same as the reference)
from ab_data import make  # noqa: E402
kind = os.environ.get("AB_DATA", "randn")
if kind == "randn":
    gen = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(n, d, generator=gen, device="cuda")
    y = torch.randn(n2, d, generator=gen, device="cuda") * 1.05 + 0.05
else:
    seed = int(os.environ.get("AB_SEED", "0"))
    x, y = make(kind, n, d, 2 * seed), make(kind, n2, d, 2 * seed + 1)
    if kind == "scales":
        y = y * 1e-3                                    # the candidate set three orders of magnitude smaller than the reference
kk = int(os.environ.get("AB_K", "5"))
rx, ry = ops.knn_radii(x, kk), ops.knn_radii(y, kk)
want_min = os.environ.get("AB_WANT_MIN", "0") == "1"
ops.prdc_counts(x, y, rx, ry, want_min)
torch.cuda.synchronize()
ts = []
for _ in range(3):
    t0 = time.perf_counter()
    outs = ops.prdc_counts(x, y, rx, ry, want_min)
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
col, rany = outs[0], outs[1]
import hashlib  # noqa: E402
digest = hashlib.sha1(b"".join(o.cpu().numpy().tobytes() for o in outs)).hexdigest()[:12]
print(f"want_min={int(want_min)} variant={os.environ.get('AM_ENGINE_VARIANT', 'default')} sha1 {digest} cross order={os.environ.get('AM_CROSS_ORDER', '0')} wg_target={os.environ.get('AM_WG_TARGET', '8192')} N={n} D={d}: "
      f"best {min(ts) * 1e3:.2f} ms {2 * n * n2 * d / min(ts) / 1e12:.1f} TF  sum {int(col.sum())} {int(rany.sum())}", flush=True)
