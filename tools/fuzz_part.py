#!/usr/bin/env python3
"""Randomised parity fuzz of the partitioned (multi-GPU) k-NN against the single-GPU entry point, one GPU emulating every
rank in turn.  Usage: tools/fuzz_part.py [n_cases] [seed]"""
import os
import random
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from audio_metrics_amd import hip_ops as ops  # noqa: E402
from ab_data import make  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(n_cases):
    rows = rnd.choice([9000, 20011, 33000, 40000, 52001, 70000])
    dim = rnd.choice([128, 136, 200, 256, 512])
    k = rnd.choice([1, 3, 5, 10])
    world = rnd.choice([2, 3, 4, 8])
    data = rnd.choice(os.environ.get("FUZZ_DATA", "randn,clustered,scales,lowrank,unit,dups,silence,hub").split(","))
    x = make(data, rows, dim, rnd.randrange(1000))
    if not ops.knn_sym_eligible(rows, dim, k):
        print(f"case {case}: rows={rows} dim={dim} k={k}: not eligible, skipped")
        continue
    want = ops.knn_radii(x, k).cpu().numpy()
    bounds_of = [(rows * p // world, rows * (p + 1) // world) for p in range(world)]
    bounds = torch.cat([ops.knn_bounds(x, k, lo, hi - lo) for lo, hi in bounds_of])
    lists = torch.stack([ops.knn_sym_part(x, k, p, world, bounds) for p in range(world)])
    got = ops.knn_lists_finish(lists, x, k).cpu().numpy()
    ok = np.array_equal(got.view(np.uint32), want.view(np.uint32))
    bad += not ok
    print(f"case {case}: rows={rows} dim={dim} k={k} world={world} data={data} path={ops.knn_path(rows, rows, dim, k)}: "
          f"{'ok' if ok else 'MISMATCH'}", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
