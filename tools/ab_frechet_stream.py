#!/usr/bin/env python3
"""where should the Frechet solve of a one-call evaluate run?  Times am_evaluate_f32 (FAD + KD + PRDC, the
bench step) with the A/B build's AM_EVAL_FAD_PLACE = 0 (side stream, free to start behind the statistics - shipped),
1 (caller's stream, behind the kernel distance), 2 (side stream, held back until the first k-NN entry has finished); prints the
step time and the mean launch time of the two tile kernels (library kernel clocks).  AB_ROWS / AB_DIM / AB_K / AB_REPS."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import inputs as gi  # noqa: E402
from audio_metrics_amd import distributed, hip_ops as ops  # noqa: E402

n, d, k = (int(os.environ.get(key, dflt)) for key, dflt in (("AB_ROWS", "100000"), ("AB_DIM", "512"), ("AB_K", "5")))
reps = int(os.environ.get("AB_REPS", "20"))
ref, cand = (torch.as_tensor(a).cuda() for a in gi.bench_pair(os.environ.get("AB_DATA", "randn"), n, d))
metrics = ("fad", "kd", "prdc")
for _ in range(3):
    res = distributed.evaluate_single(ref, cand, metrics, k, ops)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    res = distributed.evaluate_single(ref, cand, metrics, k, ops)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / reps * 1e3
# kernel clocks in a second, untimed loop (the events they record sit between the kernels)
ops.kernel_clock_enable(True)
for _ in range(5):
    distributed.evaluate_single(ref, cand, metrics, k, ops)
torch.cuda.synchronize()
clock = {name: ops.kernel_clock_read(kid) for kid, name in enumerate(("knn", "cross"))}
print(f"[{os.environ.get('AB_TAG', '-')}] place {os.environ.get('AM_EVAL_FAD_PLACE', '0')} N={n} D={d} k={k}: {ms:.3f} ms per evaluate | "
      + " ".join(f"{name} {t / max(c, 1):.3f} ms" for name, (c, t) in clock.items()) + f" | fad {res['fad']:.9f} precision {res['precision']:.6f}", flush=True)
