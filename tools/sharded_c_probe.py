#!/usr/bin/env python3
"""One rank, full bench size: the one-call-per-rank entry point am_evaluate_sharded_f32 (hooks that have nothing to exchange)
beside the fused one-GPU call am_evaluate_f32 - what the schedule's own bookkeeping costs (two row copies into the gathered
buffers, ~20 events, the hook calls).  AB_ROWS / AB_DIM."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import inputs as gi  # noqa: E402
from audio_metrics_amd import hip_ops as ops  # noqa: E402
from audio_metrics_amd.collectives import TorchCollectives  # noqa: E402
from audio_metrics_amd.metrics.kd import device_subset_indices  # noqa: E402

n, d, k = int(os.environ.get("AB_ROWS", "100000")), int(os.environ.get("AB_DIM", "512")), 5
ref, cand = (torch.as_tensor(a).cuda() for a in gi.bench_pair("randn", n, d))
i1, i2 = device_subset_indices(n, n, 100, 1000, 1234, ref.device)
what = ("fad", "kd", "prdc")


def timed(fn, reps=8):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out


def sharded(overlap):
    return ops.evaluate_sharded_c(ref, cand, [n], [n], what, TorchCollectives(None), k, i1, i2, overlap=overlap)


for overlap in (True, False, True, False):                # (the allocator settles over the first calls of either form)
    sharded(overlap)
t_fused, (h0, m0) = timed(lambda: ops.evaluate(ref, cand, what, k, i1, i2))
for overlap in (True, False, True, False):
    t_c, (h1, m1) = timed(lambda: sharded(overlap))
    same = h0[5:9] == h1[5:9] and (m0 == m1).all() and abs(h0[0] - h1[0]) <= 1e-9 * abs(h0[0])
    print(f"{n} x {d}: am_evaluate_f32 {t_fused:.2f} ms, am_evaluate_sharded_f32 (one rank, {'communication stream' if overlap else 'serial'}) "
          f"{t_c:.2f} ms, same record: {same}", flush=True)
