#!/usr/bin/env python3
"""How much of a float64 k-NN / membership call on the f16 filter route is its float64 evaluation (library kernel clocks:
knn_fast_select64_kernel under KERNEL_KNN_VERIFY, cross_verify_regions64_kernel under KERNEL_PRDC_VERIFY)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import inputs as gi  # noqa: E402
from audio_metrics_amd import hip_ops as ops  # noqa: E402

for n, d in ((100000, 64), (100000, 128), (100000, 32)):
    ref, cand = (torch.as_tensor(a).cuda() for a in gi.pair64("randn", gi.BENCH_SEED, n, n, d))
    ops.kernel_clock_enable(True)
    for rep in range(4):
        t0 = time.perf_counter()
        r_ref = ops.knn_radii(ref, 5)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        r_cand = ops.knn_radii(cand, 5)
        out = ops.prdc_counts(ref, cand, r_ref, r_cand)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if rep == 0:
            for kid in range(4):
                ops.kernel_clock_read(kid)
    clock = {name: ops.kernel_clock_read(kid) for kid, name in enumerate(("sweep", "membership filter", "knn f64 selection", "membership f64 verification"))}
    print(f"{n} x {d} float64: knn entry {(t1 - t0) * 1e3:.2f} ms, second knn + counts {(t2 - t1) * 1e3:.2f} ms | " +
          ", ".join(f"{name} {ms / max(c, 1):.3f} ms" for name, (c, ms) in clock.items()), flush=True)
