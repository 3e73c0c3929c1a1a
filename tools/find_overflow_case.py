#!/usr/bin/env python3
"""Which seeded sets send the membership filter through its overflow queue (am_filter_stats: prdc_overflow_queue) or past its
budget (prdc_fallback_calls) - how the cases of tests/test_gpu_routes.py were re-checked after round 6 changed the work-item
shape of the filter (32 row blocks per group).  Usage: python tools/find_overflow_case.py"""
import sys, os, torch
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import route_probe as rp
ops = rp.ops
ops.filter_stats_enable("cuda:0", True)
for fam, n, n2, d, k, seed in [("hub", 30000, 30000, 128, 3, 6), ("hub", 30000, 30000, 128, 3, 7), ("hub", 30000, 30000, 256, 3, 6), ("hub", 40000, 40000, 128, 3, 7),
                               ("hub", 30000, 30000, 512, 3, 6), ("hub", 30000, 30000, 512, 3, 7), ("silence", 30000, 30000, 128, 5, 6), ("silence", 40000, 40000, 256, 5, 7),
                               ("hub", 20000, 40000, 128, 3, 6), ("hub", 40000, 20000, 128, 3, 6), ("hub", 30000, 30000, 64, 3, 6), ("hub", 30000, 30000, 200, 5, 10), ("hub", 50000, 50000, 128, 3, 6)]:
    x, y = rp.make(fam, n, d, seed), rp.make(fam, n2, d, seed + 100)
    r, r2 = ops.knn_radii(x, k), ops.knn_radii(y, k)
    ops.filter_stats_read("cuda:0")
    got = ops.prdc_counts(x, y, r, r2, want_min=False)
    s = ops.filter_stats_read("cuda:0")
    print(fam, n, n2, d, k, seed, "| calls", s["prdc_calls"], "fallback", s["prdc_fallback_calls"], "overflow", s["prdc_overflow_queue"], "queued", s["prdc_queued"], flush=True)
