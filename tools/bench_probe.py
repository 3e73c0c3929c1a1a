#!/usr/bin/env python3
"""Development aid: per-step wall time of back-to-back evaluate_sharded calls (sporadic stalls show as outliers);
AB_COMBOS selects metric subsets, e.g. "fad,kd" = BASELINE configs[1]."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_metrics_amd.distributed import evaluate_sharded
n = int(os.environ.get("AB_ROWS", "20000")); d = int(os.environ.get("AB_DIM", "512"))
gen = torch.Generator(device="cuda").manual_seed(0)
ref = torch.randn(n, d, generator=gen, device="cuda"); cand = torch.randn(n, d, generator=gen, device="cuda") * 1.05 + 0.05
for combo in os.environ.get("AB_COMBOS", "fad,kd,prdc").split(";"):
    metrics = tuple(combo.split(","))
    ts = []
    for _ in range(30):
        t0 = time.perf_counter(); r = evaluate_sharded(ref, cand, metrics=metrics, nearest_k=int(os.environ.get("AB_K", "5"))); ts.append(round((time.perf_counter() - t0) * 1e3, 1))
    print(metrics, "median", sorted(ts)[15], "ms;", ts, flush=True)
