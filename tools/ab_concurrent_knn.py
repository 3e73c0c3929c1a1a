#!/usr/bin/env python3
"""Would the two k-NN entry points of an evaluate (reference set, candidate set: independent chains of a sweep and ~ten small
kernels each) finish sooner on two streams than one after the other?  Times both orders with the shipped library at the
BASELINE size; AB_ROWS / AB_K."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import inputs as gi  # noqa: E402
from audio_metrics_amd import hip_ops as ops  # noqa: E402

n, d, k = int(os.environ.get("AB_ROWS", "100000")), 512, int(os.environ.get("AB_K", "5"))
ref, cand = (torch.as_tensor(a).cuda() for a in gi.bench_pair("randn", n, d))
pr, pc = ops.prepare(ref), ops.prepare(cand)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def sequential():
    return ops.knn_radii(ref, k, prepared=pr), ops.knn_radii(cand, k, prepared=pc)


def concurrent():
    main = torch.cuda.current_stream()
    s1.wait_stream(main)
    s2.wait_stream(main)
    with torch.cuda.stream(s1):
        a = ops.knn_radii(ref, k, prepared=pr)
    with torch.cuda.stream(s2):
        b = ops.knn_radii(cand, k, prepared=pc)
    main.wait_stream(s1)
    main.wait_stream(s2)
    return a, b


for name, fn in (("sequential", sequential), ("two streams", concurrent), ("sequential", sequential), ("two streams", concurrent)):
    out = fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        out = fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    print(f"{name:12s}: {ms:.3f} ms for both sets   checksum {float(out[0].double().sum() + out[1].double().sum()):.6f}", flush=True)
