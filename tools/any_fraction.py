#!/usr/bin/env python3
"""What fraction of the reference rows has an "any" witness after a 1/16 column sample (what the membership filter's pre-pass
sees) and at the end?  Development aid for the row-ordering question in DESIGN section 8."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import inputs as gi  # noqa: E402
from audio_metrics_amd import hip_ops as ops  # noqa: E402

for kind, k in (("randn", 5), ("randn", 10), ("clap", 5), ("clap", 10)):
    ref, cand = (torch.as_tensor(a).cuda() for a in gi.bench_pair(kind, 100000, 512))
    r_ref, r_cand = ops.knn_radii(ref, k), ops.knn_radii(cand, k)
    _, rany, _ = ops.prdc_counts(ref, cand, r_ref, r_cand)
    sel = (torch.arange(cand.shape[0], device="cuda") // 128) % 16 == 0
    _, rany_s, _ = ops.prdc_counts(ref, cand[sel].contiguous(), r_ref, r_cand[sel].contiguous())
    a = rany_s.bool().view(-1)
    blocks = a[: a.numel() // 256 * 256].view(-1, 256)
    print(f"{kind} k={k}: rows with a witness: {rany.float().mean().item():.3f} at the end, {a.float().mean().item():.3f} in a 1/16 column sample; "
          f"256-row blocks entirely determined after sorting by that flag: {a.float().mean().item() * a.numel() // 256 / (a.numel() // 256):.3f}; unsorted: {blocks.all(1).float().mean().item():.4f}")
